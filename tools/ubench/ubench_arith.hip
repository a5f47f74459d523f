// Microbenchmark: which exact modular-arithmetic butterfly is cheapest on gfx950?
// Decides the NTT field for the BlindRotate hot path.
// Each variant runs NB independent butterfly chains per lane for ITERS iterations;
// we report butterflies/s over the whole chip and the implied cycles per wave-butterfly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int NB = 8;        // independent butterflies per lane
constexpr int ITERS = 2048;

// ---------------- FP64 prime (p < 2^51.x), balanced lazy residues ----------------
__device__ __forceinline__ double f64_mulmod_lazy(double a, double w, double p, double pinv)
{
    double h = a * w;
    double l = __builtin_fma(a, w, -h);
    double q = __builtin_rint(h * pinv);
    double r = __builtin_fma(-q, p, h);
    return r + l;
}
__device__ __forceinline__ double f64_mulmod_magic(double a, double w, double p, double pinv)
{
    const double M = 6755399441055744.0; // 1.5 * 2^52
    double h = a * w;
    double l = __builtin_fma(a, w, -h);
    double q = __builtin_fma(h, pinv, M) - M;
    double r = __builtin_fma(-q, p, h);
    return r + l;
}
__device__ __forceinline__ double f64_reduce(double a, double p, double pinv)
{
    double q = __builtin_rint(a * pinv);
    return __builtin_fma(-q, p, a);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_f64(double* out, const double* in, double p, double pinv)
{
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    double a[NB], b[NB], w[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) { a[i] = in[(tid * NB + i) % 4096]; b[i] = in[(tid * NB + i + 7) % 4096]; w[i] = in[(tid + i * 13) % 4096]; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NB; i++) {
            double t = (MODE == 0) ? f64_mulmod_lazy(b[i], w[i], p, pinv) : f64_mulmod_magic(b[i], w[i], p, pinv);
            double x = a[i] + t, y = a[i] - t;
            if (MODE == 2) { x = f64_reduce(x, p, pinv); }
            a[i] = y; b[i] = x;
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NB; i++) s += a[i] + b[i];
    out[tid] = s;
}

// ---------------- Goldilocks 2^64 - 2^32 + 1, u64 ----------------
constexpr uint64_t GP = 0xffffffff00000001ull;
__device__ __forceinline__ uint64_t g_add(uint64_t a, uint64_t b)
{
    uint64_t s = a + b;
    uint32_t c = s < a;
    s += (uint64_t)(0u - c);          // + (2^32-1) on carry  (0xffffffff)
    return s;                          // lazy: may be >= p, < 2^64
}
__device__ __forceinline__ uint64_t g_sub(uint64_t a, uint64_t b)
{
    uint64_t s = a - b;
    uint32_t c = a < b;
    s -= (uint64_t)(0u - c);          // - (2^32-1) on borrow
    return s;
}
__device__ __forceinline__ uint64_t g_reduce128(uint64_t lo, uint64_t hi)
{
    // x = lo + hi_lo*2^64 + hi_hi*2^96 ; 2^64 = 2^32-1 ; 2^96 = -1
    uint64_t hi_hi = hi >> 32, hi_lo = hi & 0xffffffffull;
    uint64_t t = lo - hi_hi;
    if (lo < hi_hi) t -= 0xffffffffull;
    uint64_t m = hi_lo * 0xffffffffull;  // (hi_lo<<32) - hi_lo
    uint64_t r = t + m;
    if (r < m) r += 0xffffffffull;
    return r;
}
__device__ __forceinline__ uint64_t g_mul(uint64_t a, uint64_t b)
{
    uint64_t lo = a * b;
    uint64_t hi = __umul64hi(a, b);
    return g_reduce128(lo, hi);
}
template <int SH>
__device__ __forceinline__ uint64_t g_shl(uint64_t a)   // a * 2^SH, 0 < SH < 32
{
    uint64_t lo = a << SH;
    uint64_t hi = a >> (64 - SH);      // < 2^32
    uint64_t m = hi * 0xffffffffull;
    uint64_t r = lo + m;
    if (r < m) r += 0xffffffffull;
    return r;
}
template <int MODE>
__global__ __launch_bounds__(256) void k_gold(uint64_t* out, const uint64_t* in)
{
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[NB], b[NB], w[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) { a[i] = in[(tid * NB + i) % 4096]; b[i] = in[(tid * NB + i + 7) % 4096]; w[i] = in[(tid + i * 13) % 4096]; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NB; i++) {
            uint64_t t = (MODE == 0) ? g_mul(b[i], w[i]) : g_shl<24>(b[i]);
            uint64_t x = g_add(a[i], t), y = g_sub(a[i], t);
            a[i] = y; b[i] = x;
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < NB; i++) s += a[i] ^ b[i];
    out[tid] = s;
}

// ---------------- two 31-bit primes, Shoup u32 ----------------
__device__ __forceinline__ uint32_t s_mul(uint32_t a, uint32_t w, uint32_t wq, uint32_t p)
{
    uint32_t q = __umulhi(a, wq);
    return a * w - q * p;              // in [0, 2p)
}
__global__ __launch_bounds__(256) void k_u32x2(uint32_t* out, const uint32_t* in, uint32_t p0, uint32_t p1)
{
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a0[NB], b0[NB], a1[NB], b1[NB], w0[NB], wq0[NB], w1[NB], wq1[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {
        a0[i] = in[(tid * NB + i) % 4096] >> 2; b0[i] = in[(tid * NB + i + 7) % 4096] >> 2; w0[i] = in[(tid + i * 13) % 4096] >> 2; wq0[i] = in[(tid + i * 17) % 4096];
        a1[i] = in[(tid * NB + i + 1) % 4096] >> 2; b1[i] = in[(tid * NB + i + 9) % 4096] >> 2; w1[i] = in[(tid + i * 11) % 4096] >> 2; wq1[i] = in[(tid + i * 19) % 4096];
    }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NB; i++) {
            uint32_t t = s_mul(b0[i], w0[i], wq0[i], p0);
            uint32_t x = a0[i] + t; x -= (x >= 2 * p0) ? 2 * p0 : 0;
            uint32_t y = a0[i] + 2 * p0 - t; y -= (y >= 2 * p0) ? 2 * p0 : 0;
            a0[i] = y; b0[i] = x;
            t = s_mul(b1[i], w1[i], wq1[i], p1);
            x = a1[i] + t; x -= (x >= 2 * p1) ? 2 * p1 : 0;
            y = a1[i] + 2 * p1 - t; y -= (y >= 2 * p1) ? 2 * p1 : 0;
            a1[i] = y; b1[i] = x;
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < NB; i++) s += a0[i] ^ b0[i] ^ a1[i] ^ b1[i];
    out[tid] = s;
}

// ---------------- raw instruction rates ----------------
template <int OP>
__global__ __launch_bounds__(256) void k_raw(uint32_t* out, const uint32_t* in)
{
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a[NB], b[NB];
    double d[NB], e[NB];
    uint64_t u[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) { a[i] = in[(tid + i) % 4096]; b[i] = in[(tid + i * 5 + 1) % 4096] | 1; d[i] = (double)a[i]; e[i] = 1.0 + (double)b[i] * 1e-10; u[i] = ((uint64_t)a[i] << 32) | b[i]; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NB; i++) {
            if (OP == 0) a[i] = a[i] * b[i];                       // v_mul_lo_u32
            if (OP == 1) a[i] = __umulhi(a[i], b[i]);              // v_mul_hi_u32
            if (OP == 2) u[i] = (uint64_t)(uint32_t)u[i] * b[i] + u[i]; // v_mad_u64_u32
            if (OP == 3) d[i] = __builtin_fma(d[i], e[i], e[i]);   // v_fma_f64
            if (OP == 4) d[i] = __builtin_rint(d[i]) * e[i];       // v_rndne_f64 + v_mul_f64
            if (OP == 5) a[i] = a[i] + b[i] + (a[i] >> 3);         // int add/shift
            if (OP == 6) d[i] = d[i] + e[i];                       // v_add_f64
            if (OP == 7) a[i] = __mul24(a[i] & 0xffffff, b[i] & 0xffffff) + a[i]; // v_mad_u32_u24
            if (OP == 8) u[i] = u[i] + (((uint64_t)b[i] << 32) | a[i]);      // 64-bit add
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < NB; i++) s += a[i] + (uint32_t)d[i] + (uint32_t)u[i] + (uint32_t)(u[i] >> 32);
    out[tid] = s;
}

template <class F>
static double time_ms(F f)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) f();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    const int blocks = cus * 8, threads = 256;           // 8 waves/SIMD requested
    const size_t nthr = (size_t)blocks * threads;
    std::vector<uint32_t> h(8192);
    for (auto& v : h) v = (uint32_t)rand() * 2654435761u + (uint32_t)rand();
    void *din, *dout;
    CHECK(hipMalloc(&din, 8192 * 8)); CHECK(hipMalloc(&dout, nthr * 8));
    // doubles: integers below 2^50
    std::vector<double> hd(4096);
    for (auto& v : hd) v = (double)(((uint64_t)rand() << 20 ^ rand()) & ((1ull << 50) - 1)) - (double)(1ull << 49);
    void* dind; CHECK(hipMalloc(&dind, 4096 * 8));
    CHECK(hipMemcpy(dind, hd.data(), 4096 * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(din, h.data(), 8192 * 4, hipMemcpyHostToDevice));
    const double p = 2251799813554177.0;   // some odd number near 2^51 (rate test only)
    const double pinv = 1.0 / p;
    const double bf = (double)nthr * NB * ITERS;
    auto rep = [&](const char* name, double ms, double units) {
        double rate = units / (ms * 1e-3);
        // cycles per wave-butterfly per SIMD at 2.4 GHz: SIMDs = cus*4
        double cyc = (double)cus * 4 * 2.4e9 / (rate / 64.0);
        printf("%-34s %8.3f ms  %10.3e /s  ~%7.2f SIMD-cycles per wave-op @2.4GHz\n", name, ms, rate, cyc);
    };
    rep("f64 butterfly (rint, lazy)", time_ms([&] { hipLaunchKernelGGL(k_f64<0>, blocks, threads, 0, 0, (double*)dout, (double*)dind, p, pinv); }), bf);
    rep("f64 butterfly (magic, lazy)", time_ms([&] { hipLaunchKernelGGL(k_f64<1>, blocks, threads, 0, 0, (double*)dout, (double*)dind, p, pinv); }), bf);
    rep("f64 butterfly (+1 reduce)", time_ms([&] { hipLaunchKernelGGL(k_f64<2>, blocks, threads, 0, 0, (double*)dout, (double*)dind, p, pinv); }), bf);
    rep("goldilocks butterfly (general mul)", time_ms([&] { hipLaunchKernelGGL(k_gold<0>, blocks, threads, 0, 0, (uint64_t*)dout, (uint64_t*)din); }), bf);
    rep("goldilocks butterfly (shift 24)", time_ms([&] { hipLaunchKernelGGL(k_gold<1>, blocks, threads, 0, 0, (uint64_t*)dout, (uint64_t*)din); }), bf);
    rep("2x u32 Shoup butterfly (2 primes)", time_ms([&] { hipLaunchKernelGGL(k_u32x2, blocks, threads, 0, 0, (uint32_t*)dout, (uint32_t*)din, 2013265921u, 1811939329u); }), bf);
    rep("raw v_mul_lo_u32", time_ms([&] { hipLaunchKernelGGL(k_raw<0>, blocks, threads, 0, 0, (uint32_t*)dout, (uint32_t*)din); }), bf);
    rep("raw v_mul_hi_u32", time_ms([&] { hipLaunchKernelGGL(k_raw<1>, blocks, threads, 0, 0, (uint32_t*)dout, (uint32_t*)din); }), bf);
    rep("raw v_mad_u64_u32", time_ms([&] { hipLaunchKernelGGL(k_raw<2>, blocks, threads, 0, 0, (uint32_t*)dout, (uint32_t*)din); }), bf);
    rep("raw v_fma_f64", time_ms([&] { hipLaunchKernelGGL(k_raw<3>, blocks, threads, 0, 0, (uint32_t*)dout, (uint32_t*)din); }), bf);
    rep("raw v_rndne_f64+v_mul_f64 (2 ops)", time_ms([&] { hipLaunchKernelGGL(k_raw<4>, blocks, threads, 0, 0, (uint32_t*)dout, (uint32_t*)din); }), bf);
    rep("raw int add+shift+add (3 ops)", time_ms([&] { hipLaunchKernelGGL(k_raw<5>, blocks, threads, 0, 0, (uint32_t*)dout, (uint32_t*)din); }), bf);
    rep("raw v_add_f64", time_ms([&] { hipLaunchKernelGGL(k_raw<6>, blocks, threads, 0, 0, (uint32_t*)dout, (uint32_t*)din); }), bf);
    rep("raw v_mad_u32_u24", time_ms([&] { hipLaunchKernelGGL(k_raw<7>, blocks, threads, 0, 0, (uint32_t*)dout, (uint32_t*)din); }), bf);
    rep("raw 64-bit int add (2 ops)", time_ms([&] { hipLaunchKernelGGL(k_raw<8>, blocks, threads, 0, 0, (uint32_t*)dout, (uint32_t*)din); }), bf);
    return 0;
}
