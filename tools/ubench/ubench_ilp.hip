// Microbenchmark: dependent-issue latency of FP64 VALU on gfx950.
// A wave issues ILP independent chains of dependent v_fma_f64 (the instruction order is pinned with inline asm);
// cycles per instruction by ILP and by waves per SIMD (1: 256-thread workgroups, 2: 512-thread), one workgroup per CU.
// Decides how many mulmod chains the NTT butterflies must interleave by hand (the compiler schedules them serially).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 512;
constexpr int UNROLL = 24;      // instructions per loop body (multiple of 1,2,3,4,6,8)

template <int ILP, int OP>
__global__ void k(double* out, unsigned long long* cyc, double w, double c)
{
    double x[8];
    for (int i = 0; i < 8; i++) x[i] = (double)(threadIdx.x + i);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            double& v = x[u % ILP];
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v) : "v"(w), "v"(c));
            else if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(c));
            else if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v) : "v"(w));
            else if (OP == 3) { int t; asm volatile("v_cvt_i32_f64 %0, %1\n\tv_cvt_f64_i32 %1, %0" : "=&v"(t), "+v"(v)); }      // two conversions
            else if (OP == 4) { unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
                                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi)); v = __hiloint2double((int)hi, (int)lo); }
            else if (OP == 5) { unsigned lo = (unsigned)__double2loint(v); asm volatile("v_bfe_i32 %0, %0, 3, 6" : "+v"(lo)); v = __hiloint2double(__double2hiint(v), (int)lo); }
            else { unsigned lo = (unsigned)__double2loint(v); asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(lo)); v = __hiloint2double(__double2hiint(v), (int)lo); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int ILP, int OP>
void run(const char* name, int threads)
{
    const int blocks = 256;
    double* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, blocks * threads * 8));
    CHECK(hipMalloc(&cyc, blocks * 16 * 8));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k<ILP, OP>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 0.999, 1.0);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * threads / 64);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double per = (double)h[h.size() / 2] / (ITERS * UNROLL);
    printf("%-10s ILP %d  waves/SIMD %d : %6.2f cycles per instruction per wave  (%5.2f per SIMD slot)\n", name, ILP, threads / 256, per, per / (threads / 256));
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main()
{
    for (int threads : {256, 512}) {
        run<1, 0>("v_fma_f64", threads); run<2, 0>("v_fma_f64", threads); run<3, 0>("v_fma_f64", threads); run<4, 0>("v_fma_f64", threads); run<8, 0>("v_fma_f64", threads);
        run<1, 1>("v_add_f64", threads); run<2, 1>("v_add_f64", threads); run<4, 1>("v_add_f64", threads);
        run<1, 2>("v_mul_f64", threads); run<2, 2>("v_mul_f64", threads); run<4, 2>("v_mul_f64", threads);
        run<1, 3>("2 x cvt", threads); run<4, 3>("2 x cvt", threads);
        run<1, 4>("permlane32", threads); run<4, 4>("permlane32", threads);
        run<1, 5>("v_bfe_i32", threads); run<4, 5>("v_bfe_i32", threads);
        run<1, 6>("v_add_u32", threads); run<4, 6>("v_add_u32", threads);
    }
    return 0;
}
