// Microbenchmark: is the matrix pipe free FP64 issue beside a saturated vector pipe?  (gfx950 / MI355X)
//
// The blind rotation is bound by FP64 VECTOR issue (v_fma_f64 / v_mul_f64 / v_add_f64 at one wave-instruction per 4 cycles per
// SIMD, pipe 0.80 busy) while the matrix cores idle.  v_mfma_f64_16x16x4_f64 multiplies exactly as long as products and 4-term sums
// stay below 2^53: a 6-bit gadget digit times a 25-bit twiddle limb does.  Before touching the kernel: how many such MFMAs does a
// SIMD retire BESIDE a saturated v_fma_f64 stream at the kernel's occupancy (2 waves per SIMD, 8 waves per workgroup, one workgroup
// per CU)?  Three instruction streams, order pinned with inline asm:
//   V   16 independent v_fma_f64 chains (the saturated vector stream)
//   M   K independent accumulator tiles of v_mfma_f64_16x16x4_f64
//   VM  per loop body 16 v_fma_f64 with m MFMAs interleaved (m = 1, 2, 4)
// Reported per SIMD: cycles per loop body, v_fma_f64 per cycle, exact multiply-accumulates per cycle of each pipe, and for VM the
// slowdown of the vector stream against V.  Shader cycles from s_memtime; one launch of 256 x 512 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2048;
constexpr int VFMA = 16;       // vector FMAs per loop body

// MODE 0: V, 1: M (MF tiles per body), 2: VM (MF MFMAs spread among the 16 FMAs)
template <int MODE, int MF>
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, double w, double c)
{
    double x[VFMA];
    v4f64 acc[4];
    for (int i = 0; i < VFMA; i++) x[i] = (double)(threadIdx.x + i) * 1e-3;
    for (int i = 0; i < 4; i++) acc[i] = v4f64{0.0, 0.0, 0.0, 0.0};
    const double a = (double)(threadIdx.x & 63), b = (double)(threadIdx.x >> 6) + 1.0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < VFMA; u++) {
            if (MODE != 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[u]) : "v"(w), "v"(c));
            if (MODE != 0 && MF > 0 && (u % (VFMA / MF)) == 0) {
                v4f64& t = acc[(u / (VFMA / MF)) % 4];
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(t) : "v"(a), "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < VFMA; i++) s += x[i];
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

struct Row { const char* name; int vfma, mfma; double cycles; };

template <int MODE, int MF>
Row run(const char* name, double* d_out, unsigned long long* d_cyc, int blocks)
{
    const int waves = blocks * 8;
    std::vector<unsigned long long> h(waves);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((k<MODE, MF>), dim3(blocks), dim3(512), 0, 0, d_out, d_cyc, 0.999999, 1e-9);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), d_cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    // s_memtime counts the constant 100 MHz clock on this chip: convert with the shader clock measured by the V stream (known issue rate)
    return {name, MODE == 1 ? 0 : VFMA, MODE == 0 ? 0 : MF, (double)h[waves / 2] / ITERS};
}

int main()
{
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int blocks = prop.multiProcessorCount;
    double* d_out;
    unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_out, (size_t)blocks * 512 * sizeof(double)));
    CHECK(hipMalloc(&d_cyc, (size_t)blocks * 8 * sizeof(unsigned long long)));
    std::vector<Row> rows;
    rows.push_back(run<0, 0>("V   16 v_fma_f64", d_out, d_cyc, blocks));
    rows.push_back(run<1, 1>("M   1 mfma", d_out, d_cyc, blocks));
    rows.push_back(run<1, 2>("M   2 mfma", d_out, d_cyc, blocks));
    rows.push_back(run<1, 4>("M   4 mfma", d_out, d_cyc, blocks));
    rows.push_back(run<2, 1>("VM  16 v_fma_f64 + 1 mfma", d_out, d_cyc, blocks));
    rows.push_back(run<2, 2>("VM  16 v_fma_f64 + 2 mfma", d_out, d_cyc, blocks));
    rows.push_back(run<2, 4>("VM  16 v_fma_f64 + 4 mfma", d_out, d_cyc, blocks));
    // Units: the counter ticks are whatever s_memtime counts; everything is reported RELATIVE to the V stream, whose rate is known
    // (one v_fma_f64 wave-instruction per 4 shader cycles per SIMD when saturated: 2 waves x 16 instructions = 128 cycles per body).
    const double tick_per_body_V = rows[0].cycles;
    const double cycles_per_tick = 128.0 / tick_per_body_V;
    printf("%s, %d CUs, 8 waves per workgroup (2 per SIMD), %d loop bodies; cycles = shader cycles per loop body per SIMD (both waves),\n"
           "calibrated on the V stream (2 x 16 v_fma_f64 at 4 cycles each = 128)\n", prop.name, blocks, ITERS);
    printf("%-30s %10s %14s %22s %22s %12s\n", "stream", "cycles", "v_fma/cycle", "vector MACs/cycle/SIMD", "matrix MACs/cycle/SIMD", "V slowdown");
    for (const Row& r : rows) {
        const double cyc = r.cycles * cycles_per_tick;
        const double vf = 2.0 * r.vfma / cyc, vmac = vf * 64.0, mmac = 2.0 * r.mfma * 1024.0 / cyc;      // 64 lanes per v_fma; 16 x 16 x 4 MACs per MFMA
        printf("%-30s %10.1f %14.3f %22.1f %22.1f %12.3f\n", r.name, cyc, vf, vmac, mmac, r.vfma ? cyc / 128.0 : 0.0);
    }
    printf("\nreading: a row VM whose 'V slowdown' stays near 1.0 retires its MFMAs for free; 'matrix MACs' beside 'vector MACs' of row V (16.0) is the extra\n"
           "exact multiply-accumulate rate the matrix pipe would add -- before the limb split (a 50-bit twiddle = two 25-bit limbs: two MFMAs per\n"
           "product) and the recombination of the limbs mod p on the vector pipe (>= 8 v_*_f64 per output element).\n");
    return 0;
}
