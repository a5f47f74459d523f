// kernels_lvl2.hip.h -- the gate path over the N = 2048 ring with a 64-bit torus
// (BASELINE.json configs[4]; parameter set lvl2 / lvl02 / lvl20 of SURVEY.md appendix C).
//
// The reference has no N = 2048 code and no 64-bit accumulate: what is
// built here is what its gate templates compute when instantiated at brP = lvl02,
// iksP = lvl20 (include/gatebootstrapping_gpu.cuh:10-52,115-345, src/bootstrap_gpu.cu:366-421,
// include/keyswitch_gpu.cuh:83-134), with 64-bit decomposition constants.
//
// Exactness.  The external product is an exact integer sum mod 2^64.  A key word w (signed
// 64 bit) is split into three balanced limbs w = w0 + 2^22 w1 + 2^44 w2, |w0|,|w1| <= 2^21,
// |w2| <= 2^19, and the product is taken limb by limb over the FP64 prime of fpfield.h:
// |sum| <= (k+1) l N (Bg/2) 2^21 = 8 * 2048 * 256 * 2^21 = 2^43 < p/2.  The three exact sums
// are recombined with shifts mod 2^64.  The gadget digits are transformed once and shared
// by the three limbs.
//
// Transform.  X^2048 + 1 = (X^1024 - I)(X^1024 + I) with I = psi^1024 the 25-bit fourth
// root: the first forward stage (a, b) -> (a + I b, a - I b) is exact in FP64 for gadget
// digits and for key limbs, and leaves two independent 1024-point transforms that differ
// from the lvl1 transform of ntt_wave.h only in their twiddle tables (root_h[m + g] =
// root[2m + h m + g]).  One wave therefore runs a half with the lvl1 code and 16 registers
// per lane; the two halves of a row are processed one after the other.
//
// Work split.  One 8-wave workgroup per blind rotation.  Wave w owns TRGSW row w = (j, d) and the pairs (e, e + 1024),
// e = 256 d + lane + 64 m, of accumulator component j -- in registers.  Per CMux step: the owners decompose their pairs
// (rotated operand from an LDS copy of the accumulator) and hand digit d to wave (j, d) through its transpose tile; every
// wave transforms its row and adds the products with its 6 key polynomials (2 outputs x 3 limbs) into the sums of that
// half with ds_add_f64, both halves back to back; the twelve inverse half-transforms run three per SIMD in one phase;
// every wave recombines the halves and the three limbs of its pairs into its accumulator registers.  Five workgroup
// barriers per step.  LDS map and schedule: at the kernel.
#pragma once
#include "kernels.hip.h"

namespace cufhe_amd {

constexpr int k2Nbit = 11;
constexpr int k2N = 2048;              // lvl2param::n
constexpr int k2L = 4;                 // lvl2param::l
constexpr int k2Bgbit = 9;             // lvl2param::Bgbit
constexpr int k2KsT = 7;               // lvl20param::t
constexpr int k2KsBasebit = 2;         // lvl20param::basebit
constexpr uint64_t k2Mu = 1ull << 61;  // lvl2param::mu
constexpr int k2Words = k2N + 1;       // words of a lvl2 TLWE
constexpr int k2BkRows = 2 * k2L;      // 8
constexpr int k2Limbs = 3;
constexpr int k2LimbBits = 22;
constexpr int k2Prods = 2 * k2Limbs;   // key polynomials per row: (out, limb)
constexpr int k2Half = k2N / 2;        // 1024 = kN: a half transform is a lvl1-sized transform
static_assert(k2Half == kN, "the half transforms reuse ntt_wave.h");
// NTT-domain key: [step][half][row][out * 3 + limb][1024] doubles, each polynomial in the
// layout-C order of bk_to_ntt_kernel
constexpr size_t k2BkStepDoubles = (size_t)2 * k2BkRows * k2Prods * k2Half;   // 98304 = 768 KiB
constexpr int k2KsNumBase = 1 << (k2KsBasebit - 1);

// exactness per limb: |sum| <= (k+1) l N (Bg/2) 2^(limb bits - 1) < p/2; the top limb holds 64 - 2*22 = 20 bits
constexpr double k2LimbSumBound = 2.0 * k2L * k2N * (double)(1u << (k2Bgbit - 1)) * (double)(1u << (k2LimbBits - 1));
static_assert(k2LimbSumBound < fpf::P / 2, "lvl2 limb products do not fit the FP64 prime: use narrower limbs");
static_assert(k2Limbs * k2LimbBits >= 64, "limbs do not cover the 64-bit key word");
static_assert(k2L == 4 && k2N == 4 * 512, "the decomposition is shared by the l = 4 waves of a row group, 256 pairs (e, e + 1024) each");
// digits of magnitude Bg/2 through the first (exact) split stage and a full reducing half transform,
// reduced before the products: each product <= 0.5 + 0.0973 * 0.5, 2 l rows accumulate in LDS
static_assert((double)(1u << (k2Bgbit - 1)) * (1.0 + fpf::ROOT4) < 9007199254740992.0 / 1024.0, "first split stage is not exact");
static_assert(k2BkRows * fpf::after_mulmod(0.5001) < fpf::LIM_WIDE, "lvl2 row sums exceed 2^53");
// half 0, stage 0 without reduction: (Bg/2) (1 + I) (1 + zeta) = 2^45.2 stays an exact double with 7 bits to spare, 0.05 p
static_assert((double)(1u << (k2Bgbit - 1)) * (1.0 + fpf::ROOT4) * (1.0 + fpf::ROOT8) < 9007199254740992.0 / 128.0, "exact stage 0 of half 0");

struct RotDesc2 {          // lvl0 operands, lvl2 result (sample-extracted TLWE)
    const uint32_t* in0;
    const uint32_t* in1;
    uint64_t* out;
    int32_t ca, cb;
    uint32_t off;
    uint32_t pad;
};
struct LinDesc64 {         // out(lvl0) = KS(ca * in0 + cb * in1 + (0, .., off)) on lvl2 TLWEs
    const uint64_t* in0;
    const uint64_t* in1;
    uint32_t* out;
    int32_t ca, cb;
    uint64_t off;
};

__host__ __device__ constexpr uint64_t decomp_offset2()
{
    uint64_t o = 0;
    for (int i = 1; i <= k2L; i++) o += (1ull << (k2Bgbit - 1)) << (64 - i * k2Bgbit);
    return o + (1ull << (64 - k2L * k2Bgbit - 1));     // + roundoffset
}
__host__ __device__ constexpr uint64_t decomp_signmask2()
{
    uint64_t m = 0;
    for (int i = 1; i <= k2L; i++) m |= (1ull << (k2Bgbit - 1)) << (64 - i * k2Bgbit);
    return m;
}

// wave context whose per-lane twiddle tables live in global memory (tables of half `h`)
__device__ __forceinline__ WaveCtx make_wave_ctx_gtab(char* lds, int tile_off, const NttTables* gt, int lane)
{
    const int lam = lane & 15, hi = lane >> 4;
    WaveCtx c;
    c.a65 = lds + opaque(tile_off + 8 * lane);
    c.a66 = c.a65;
    c.b65 = lds + opaque(tile_off + 8 * (65 * lam + hi));
    c.b66 = lds + opaque(tile_off + 8 * (66 * lam + hi));
    c.tb_fwd = (const char*)gt->tb_fwd + 8 * lam;
    c.tb_inv = (const char*)gt->tb_inv + 8 * lam;
    c.tc_fwd = (const char*)gt->tc_fwd + 8 * lane;
    c.tc_inv = (const char*)gt->tc_inv + 8 * lane;
    c.gt = gt;
    return c;
}

// integer value (mod 2^64) of an integer-valued double |c| < 2^51
__device__ __forceinline__ uint64_t to_u64(double c)
{
    const double t = c + fpf::MAGIC0;                  // mantissa = 2^51 + c
    uint64_t bits;
    __builtin_memcpy(&bits, &t, 8);
    return (bits & ((1ull << 52) - 1)) - (1ull << 51);
}

// ----------------------------------------------------------------------------------
// BK (torus, uint64) -> NTT domain.  One wave per (polynomial, limb).
// bk: [step][row][out][2048]; bk_ntt as described above, scaled by 2048^-1.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kNttThreads) void bk2_to_ntt_kernel(
    double* __restrict__ bk_ntt, const uint64_t* __restrict__ bk, size_t polys,
    const NttTables* __restrict__ gt2, double n_inverse)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * kNttWavesPerBlock + wave;
    if (w >= polys * k2Limbs) return;
    const size_t poly = w / k2Limbs;
    const int limb = (int)(w % k2Limbs);
    const size_t step = poly / (2 * k2BkRows);
    const int row = (int)((poly / 2) % k2BkRows), out = (int)(poly % 2);
    const uint64_t* src = bk + poly * k2N;
    double x0[kRegs], x1[kRegs];
#pragma unroll
    for (int r = 0; r < kRegs; r++) {
        double v[2];
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            int64_t s = (int64_t)src[hh * k2Half + lane + 64 * r];
            int64_t l = (s << (64 - k2LimbBits)) >> (64 - k2LimbBits);
            for (int m = 0; m < limb; m++) {
                s = (s - l) >> k2LimbBits;
                l = (m + 1 == k2Limbs - 1) ? s : (s << (64 - k2LimbBits)) >> (64 - k2LimbBits);
            }
            v[hh] = (double)l;
        }
        x0[r] = __builtin_fma(v[1], fpf::ROOT4, v[0]);      // exact: |I b| < 2^47
        x1[r] = __builtin_fma(-v[1], fpf::ROOT4, v[0]);
    }
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        const WaveCtx ctx = make_wave_ctx_gtab(smem, wave * kTileBytes, gt2 + h, lane);
        double x[kRegs];
#pragma unroll
        for (int r = 0; r < kRegs; r++) x[r] = h ? x1[r] : x0[r];
        ntt_forward<false>(x, ctx);
        double2* dst = (double2*)(bk_ntt + (((step * 2 + h) * k2BkRows + row) * k2Prods + out * k2Limbs + limb) * k2Half);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            double2 v;
            v.x = fpf::reduce(fpf::mulmod_wide(x[2 * q], n_inverse));
            v.y = fpf::reduce(fpf::mulmod_wide(x[2 * q + 1], n_inverse));
            dst[q * 64 + lane] = v;
        }
    }
}

// ----------------------------------------------------------------------------------
// Blind rotate lvl02 + sample extract, one 8-wave workgroup per rotation.
// ----------------------------------------------------------------------------------
constexpr int k2Threads = 512;
constexpr int k2TbBytes = 2 * kTbCount * 16 * 8;                 // stage 4-7 twiddles of one half, forward and inverse: 3840

__device__ __forceinline__ void load_key_poly(double2 (&b)[8], const double* poly, int lane)
{
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_NO_BK)       // timing-only: no key loads
#pragma unroll
    for (int q = 0; q < 8; q++) b[q] = make_double2(1234567.0 + q + (double)(uintptr_t)poly * 1e-30, 7654321.0 + lane);
    return;
#endif
    const double2* p = (const double2*)poly;
#pragma unroll
    for (int q = 0; q < 8; q++) b[q] = p[q * 64 + lane];
}
// sums[reg][lane] += x * key, x reduced (|x| <= p/2): each product is below 0.55 p, eight
// rows stay below 4.4 p
__device__ __forceinline__ void accumulate_poly(double* sums, const double (&x)[kRegs], const double2 (&b)[8], int lane)
{
    double* s0 = sums + lane;
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_L2_NOATOM)       // timing-only: plain stores instead of LDS atomics
#pragma unroll
    for (int q = 0; q < 8; q++) {
        s0[(2 * q) * 64] = fpf::mulmod(x[2 * q], b[q].x);
        s0[(2 * q + 1) * 64] = fpf::mulmod(x[2 * q + 1], b[q].y);
    }
    return;
#endif
#pragma unroll
    for (int q = 0; q < 8; q++) {
        __hip_atomic_fetch_add(s0 + (2 * q) * 64, fpf::mulmod(x[2 * q], b[q].x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(s0 + (2 * q + 1) * 64, fpf::mulmod(x[2 * q + 1], b[q].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// ----------------------------------------------------------------------------------
// The schedule (the kernel it replaces -- one sum region reused by the two halves, accumulator in LDS, 64-bit LDS atomics,
// inverse jobs 8 + 4 -- is in the history of this file; its per-phase cycle counters are
// profiles/r03_lvl2_phases_round2_kernel.txt).
//
// What those counters showed (tools/lvl2_phases.py):
// the two waves of a SIMD run the same phase, the older one is served first, finishes its forward transform and products
// while the younger one transforms, and then waits at the barrier while the younger one runs its products ALONE -- a phase
// bound by the CU's vector-memory path (768 KiB of key per step at 64 B/clk = 12.3 k of the step's 56.8 k cycles; without key
// loads the step takes 48.0 k, without LDS atomics as long as with them).  The barrier between the halves (the six sums were
// one region, reused) re-aligned the waves twice per step, and the twelve inverse jobs left SIMDs half empty.  Here:
//   * the accumulator lives in REGISTERS (wave (j, c) owns the pairs (e, e + 1024), e = 256 c + lane + 64 m, of acc_j: the
//     pairs it decomposes) and the transposes go through half-size tiles (ntt_wave.h: xpose_half_tile), which frees 65 KiB
//     of LDS: the sums of the two halves get a region each, so a wave runs forward-products-forward-products without a
//     barrier and the two waves of a SIMD fall into anti-phase by themselves (one transforms while the other streams key);
//   * the twelve inverse jobs run in ONE phase, three per SIMD: waves 0-3 (served first by their SIMDs) take two sums of
//     one half each, stage by stage in one instruction stream with the twiddles loaded once, waves 4-7 one sum each; the
//     results stay where the sums were, and after one barrier every wave recombines its pairs of its output (last stage,
//     lift, the three limbs shifted and summed in registers: no 64-bit LDS atomics);
//   * the first key polynomials of the next step are requested behind the inverse jobs, when nothing else uses the
//     vector-memory path;
//   * the rotated read of the decomposition needs the accumulator in LDS: the owners leave a copy in sum regions whose
//     slice only they read, zeroed again after the rotated reads.
// Five workgroup barriers per step (six before).
// LDS: 8 half tiles 33 KiB, sums 2 x 48 KiB, abar list, stage 4-7 twiddles 7.5 KiB, stage 8-9 forward twiddles 12 KiB
// (the inverse ones are read out of them, mirrored and negated), stage 0-3 twiddles 0.5 KiB = 153 872 B.
// ----------------------------------------------------------------------------------
constexpr int k3LdsTiles = 0;
constexpr int k3LdsSum = k3LdsTiles + 8 * kHalfTileBytes;        // 33792: [half][out * 3 + limb][1024] f64
constexpr int k3SumDoubles = k2Prods * k2Half;                   // 6144 per half
constexpr int k3LdsAbar = k3LdsSum + 2 * k3SumDoubles * 8;       // + 98304
constexpr int k3LdsTb = k3LdsAbar + kAbarBytes + 16;
constexpr int k3LdsTc = k3LdsTb + 2 * k2TbBytes;                 // stage 8-9 forward twiddles of both halves: [h][12][64]
constexpr int k3TcBytes = kTcCount * 64 * 8;                     // 6144 per half
constexpr int k3LdsTu = k3LdsTc + 2 * k3TcBytes;                 // stage 0-3 twiddles: [h][tu_fwd[16] | tu_inv[16]]
constexpr int k3LdsBytes = k3LdsTu + 2 * 256;                    // 153872
static_assert(k3LdsBytes <= 160 * 1024, "lvl2 blind rotate does not fit the CU's LDS");
static_assert(2 * k2Half * 8 <= 2 * k2Half * 8 && 2 * k2N * 8 <= 2 * 2 * k2Half * 8, "accumulator copy fits two sum polynomials per component");

__global__ __launch_bounds__(k2Threads) void blind_rotate_lvl2_kernel(
    const RotDesc2* __restrict__ descs, int count, const double* __restrict__ bk_ntt,
    const NttTables* __restrict__ gt2, int steps, uint64_t* __restrict__ acc_dump)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = blockIdx.x;
    if (g >= count) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    double* sumL = (double*)(smem + k3LdsSum);                // [h][out * 3 + limb][reg][lane]
    uint16_t* abar_lds = (uint16_t*)(smem + k3LdsAbar);
    uint32_t* bbar_slot = (uint32_t*)(smem + k3LdsAbar + kAbarBytes);

    const RotDesc2 d = descs[g];
    for (int i = tid; i <= kLvl0N; i += k2Threads) {
        const uint32_t c = (uint32_t)d.ca * d.in0[i] + (uint32_t)d.cb * d.in1[i];
        if (i < kLvl0N) abar_lds[i] = (uint16_t)((c + (1u << (32 - 2 - k2Nbit))) >> (32 - 1 - k2Nbit));
        else *bbar_slot = 2 * k2N - ((c + d.off) >> (32 - 1 - k2Nbit));
    }
    for (int i = tid; i < 2 * k3SumDoubles; i += k2Threads) sumL[i] = 0.0;
    for (int i = tid; i < 2 * 2 * kTbCount * 16; i += k2Threads) {     // tb_fwd and tb_inv are contiguous in NttTables
        const int h = i / (2 * kTbCount * 16), k = i % (2 * kTbCount * 16);
        ((double*)(smem + k3LdsTb))[i] = gt2[h].tb_fwd[k];
    }
    for (int i = tid; i < 2 * kTcCount * 64; i += k2Threads) {
        const int h = i / (kTcCount * 64), k = i % (kTcCount * 64);
        ((double*)(smem + k3LdsTc))[i] = gt2[h].tc_fwd[k];
    }
    if (tid < 64) ((double*)(smem + k3LdsTu))[tid] = gt2[tid >> 5].tu_fwd[tid & 31];     // tu_fwd[16], tu_inv[16] are contiguous
    __syncthreads();

    const int wj = wave / k2L, wd = wave % k2L;               // TRGSW row (wj, wd); owner of acc_wj at e = 256 wd + lane + 64 m (+ 1024)
    const int e_first = 256 * wd + lane;
    // accumulator: RotatedTestVector<lvl2param>, include/gatebootstrapping_gpu.cuh:29-52
    uint64_t acc_lo[4], acc_hi[4];
    {
        const uint32_t bbar = *bbar_slot;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint32_t e0 = (uint32_t)(e_first + 64 * m), e1 = e0 + k2Half;
            const bool n0 = (bbar != 2 * k2N) && ((e0 < (bbar & (k2N - 1))) != ((bbar >> k2Nbit) != 0));
            const bool n1 = (bbar != 2 * k2N) && ((e1 < (bbar & (k2N - 1))) != ((bbar >> k2Nbit) != 0));
            acc_lo[m] = wj ? (n0 ? 0ull - k2Mu : k2Mu) : 0ull;
            acc_hi[m] = wj ? (n1 ? 0ull - k2Mu : k2Mu) : 0ull;
        }
    }
    // the LDS copy of acc_j for the rotated reads: sum polynomials (h, o) = (j, 3 j) and (j, 3 j + 1) -- of those the slice
    // a wave writes is read, in the recombination, by that wave only
    char* stage = smem + k3LdsSum + (wj * k3SumDoubles + 3 * wj * k2Half) * 8;
    auto publish_acc = [&]() {
        char* o = stage + opaque(8 * e_first);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            *(uint64_t*)(o + 512 * m) = acc_lo[m];
            *(uint64_t*)(o + 512 * m + 8 * k2Half) = acc_hi[m];
        }
    };
    WaveCtx ctx_tile;
    {
        const int lam8 = lane & 7, hi = lane >> 4;
        const int tile_off = k3LdsTiles + wave * kHalfTileBytes;
        ctx_tile.a65 = smem + opaque(tile_off + 8 * lane);
        ctx_tile.a66 = ctx_tile.a65;
        ctx_tile.b65 = smem + opaque(tile_off + 8 * (65 * lam8 + hi));
        ctx_tile.b66 = smem + opaque(tile_off + 8 * (66 * lam8 + hi));
        ctx_tile.tb_fwd = ctx_tile.tb_inv = ctx_tile.tc_fwd = ctx_tile.tc_inv = nullptr;
        ctx_tile.gt = gt2;
    }
    // table addresses of half hh, rebuilt where needed (a few address adds) instead of being carried through the step
    auto half_ctx = [&](int hh) {
        WaveCtx c = ctx_tile;
        int o16 = 8 * (lane & 15), o64 = 8 * lane;
        asm volatile("" : "+v"(o16), "+v"(o64));
        const NttTables* gth = gt2 + hh;
        c.tb_fwd = smem + (k3LdsTb + hh * k2TbBytes) + o16;
        c.tb_inv = c.tb_fwd + 8 * kTbCount * 16;
        c.tc_fwd = smem + (k3LdsTc + hh * k3TcBytes) + o64;
        c.tc_inv = (const char*)gth->tc_inv + o64;
        c.gt = gth;
        c.tu_l = smem + (k3LdsTu + hh * 256);
        return c;
    };
    // inverse half-transform of sum (o, hh), left in place in natural order.  Its stage 9-8 twiddles are not stored: with
    // root[i] = psi^bitrev(i) and psi^2048 = -1, root^-1[M + G] = -root[M + (M - 1 - G)], which for the half tables reads
    // inv_h[m + g] = -fwd_{1-h}[m + (m - 1 - g)]: the inverse twiddle of lane L is minus the FORWARD twiddle of the other
    // half at lane 63 - L, levels in reverse order -- read from the forward tables already in LDS (the sign rides on the
    // multiplications' source modifiers).
    double twc[kTcCount];
    auto load_twc = [&](int hh) {
        int o63 = 8 * (63 - lane);
        asm volatile("" : "+v"(o63));
        const char* t = smem + (k3LdsTc + (1 - hh) * k3TcBytes) + o63;
#pragma unroll
        for (int k = 0; k < kTcCount; k++) twc[k] = -lds_ld(t, 512 * (k < 4 ? 3 - k : 15 - k));
    };
    auto inverse_job = [&](int o, int hh) {
        load_twc(hh);
        const WaveCtx ctx = half_ctx(hh);
        double* s = sumL + hh * k3SumDoubles + o * k2Half + lane;
        double A[kRegs];
#pragma unroll
        for (int r = 0; r < kRegs; r++) A[r] = fpf::reduce(s[r * 64]);
        ntt_inverse_twc<true, true>(A, ctx, twc);           // |A| <= 2 p, natural order
#pragma unroll
        for (int r = 0; r < kRegs; r++) s[r * 64] = A[r];
    };
    // two sums of one half in one wave
    auto inverse_job2 = [&](int o1, int o2, int hh) {
        load_twc(hh);
        const WaveCtx ctx = half_ctx(hh);
        double* s1 = sumL + hh * k3SumDoubles + o1 * k2Half + lane;
        double* s2 = sumL + hh * k3SumDoubles + o2 * k2Half + lane;
        double A[kRegs], B[kRegs];
#pragma unroll
        for (int r = 0; r < kRegs; r++) { A[r] = fpf::reduce(s1[r * 64]); B[r] = fpf::reduce(s2[r * 64]); }
        ntt_inverse2_twc<true, true>(A, B, ctx, twc);
#pragma unroll
        for (int r = 0; r < kRegs; r++) { s1[r * 64] = A[r]; s2[r * 64] = B[r]; }
    };
    // last inverse stage (a, b) -> (a + b, (a - b) I^-1), I^-1 = -I, of the three limbs of output wj at this wave's pairs,
    // centred lift, limbs shifted and summed into the accumulator registers; the sums are left zero for the next step
    auto recombine = [&]() {
        double* s0 = sumL + wj * k2Limbs * k2Half + e_first;
        double u0[k2Limbs][4], u1[k2Limbs][4];
#pragma unroll
        for (int l = 0; l < k2Limbs; l++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                u0[l][m] = s0[l * k2Half + 64 * m];
                u1[l][m] = s0[k3SumDoubles + l * k2Half + 64 * m];
            }
#pragma unroll
        for (int l = 0; l < k2Limbs; l++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                s0[l * k2Half + 64 * m] = 0.0;
                s0[k3SumDoubles + l * k2Half + 64 * m] = 0.0;
                const double lo = fpf::reduce(u0[l][m] + u1[l][m]);
                const double hi = fpf::reduce(fpf::mulmod(u0[l][m] - u1[l][m], -fpf::ROOT4));
                acc_lo[m] += to_u64(lo) << (k2LimbBits * l);
                acc_hi[m] += to_u64(hi) << (k2LimbBits * l);
            }
    };

    double2 kb[3][8];
#define CUFHE_AMD_KEYPOLY3(key, k) ((key) + (size_t)((k) / k2Prods) * (k2BkRows * k2Prods * k2Half) + ((k) % k2Prods) * k2Half)
    auto prefetch_key = [&](int step) {
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_BK0)
        const double* key = bk_ntt + (size_t)(step & 1) * k2BkStepDoubles + (size_t)wave * (k2Prods * k2Half);
#else
        const double* key = bk_ntt + (size_t)step * k2BkStepDoubles + (size_t)wave * (k2Prods * k2Half);
#endif
        load_key_poly(kb[0], CUFHE_AMD_KEYPOLY3(key, 0), lane);
        load_key_poly(kb[1], CUFHE_AMD_KEYPOLY3(key, 1), lane);
    };
    if (steps > 0) prefetch_key(0);
    publish_acc();
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_PHASES)
    // timing-only: cycles of this wave per phase: [0] copy visible (barrier), [1] rotated reads + barrier, [2] digits + barrier,
    // [3] fwd h0, [4] prod h0, [5] fwd h1, [6] prod h1, [7] barrier, [8] inverse jobs, [9] barrier, [12] recombination
    unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#define CUFHE_AMD_PHASE3(k) { const unsigned long long tn = __builtin_readcyclecounter(); ph[k] += tn - tc; tc = tn; }
#else
#define CUFHE_AMD_PHASE3(k)
#endif
#pragma unroll 1
    for (int i = 0; i < steps; i++) {
        __syncthreads();                                      // every owner's copy of the accumulator is in LDS
        CUFHE_AMD_PHASE3(0)
        const uint32_t abar = __builtin_amdgcn_readfirstlane((uint32_t)abar_lds[i]);
        const int alo = (int)(abar & (k2N - 1));
        const bool ahi = (abar >> k2Nbit) != 0;
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_BK0)
        const double* key = bk_ntt + (size_t)(i & 1) * k2BkStepDoubles + (size_t)wave * (k2Prods * k2Half);
#else
        const double* key = bk_ntt + (size_t)i * k2BkStepDoubles + (size_t)wave * (k2Prods * k2Half);
#endif
        // Decomposition of (X^abar - 1) acc_wj at this wave's pairs, all four digits; digit d, packed per pair, goes to the
        // (idle) transpose tile of wave (wj, d).
        uint32_t ab[kRegs];
        {
            uint64_t rot0[4], rot1[4];
            const int rb = (e_first - alo) & (k2N - 1);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int i0 = (rb + 64 * m) & (k2N - 1);
                rot0[m] = *(const uint64_t*)(stage + 8 * i0);
                rot1[m] = *(const uint64_t*)(stage + 8 * (i0 ^ k2Half));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();                                  // all rotated reads done: the copy is dead
            CUFHE_AMD_PHASE3(1)
            {   // the two sum polynomials the copy occupied are zero again before any product is added (barrier below)
                char* z = smem + k3LdsSum + ((wave >> 2) * k3SumDoubles + 3 * (wave >> 2) * k2Half) * 8 + opaque((wave & 3) * 4096 + 16 * lane);
#pragma unroll
                for (int t = 0; t < 4; t++) *(double2*)(z + 1024 * t) = make_double2(0.0, 0.0);
            }
            char* dig_out = smem + opaque(k3LdsTiles + (wj * k2L) * kHalfTileBytes + 4 * e_first);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int e0 = e_first + 64 * m;
                const bool neg0 = (e0 < alo) != ahi, neg1 = (e0 + k2Half < alo) != ahi;
                const uint64_t t0 = ((neg0 ? 0ull - rot0[m] : rot0[m]) - acc_lo[m] + decomp_offset2()) ^ decomp_signmask2();
                const uint64_t t1 = ((neg1 ? 0ull - rot1[m] : rot1[m]) - acc_hi[m] + decomp_offset2()) ^ decomp_signmask2();
#pragma unroll
                for (int dd = 0; dd < k2L; dd++) {
                    constexpr int kTop = 64 - k2Bgbit;
                    const int pos = kTop - k2Bgbit * dd;
                    uint32_t a, b;
                    if (pos >= 32) {
                        a = (uint32_t)__builtin_amdgcn_sbfe((uint32_t)(t0 >> 32), (uint32_t)(pos - 32), (uint32_t)k2Bgbit);
                        b = (uint32_t)__builtin_amdgcn_sbfe((uint32_t)(t1 >> 32), (uint32_t)(pos - 32), (uint32_t)k2Bgbit);
                    } else {
                        a = (uint32_t)__builtin_amdgcn_sbfe(__builtin_amdgcn_alignbit((uint32_t)(t0 >> 32), (uint32_t)t0, (uint32_t)pos), 0u, (uint32_t)k2Bgbit);
                        b = (uint32_t)__builtin_amdgcn_sbfe(__builtin_amdgcn_alignbit((uint32_t)(t1 >> 32), (uint32_t)t1, (uint32_t)pos), 0u, (uint32_t)k2Bgbit);
                    }
                    *(uint32_t*)(dig_out + dd * kHalfTileBytes + 256 * m) = __builtin_amdgcn_perm(b, a, 0x05040100u);   // (a & 0xffff) | (b << 16)
                }
            }
            __syncthreads();
            const char* dig_in = smem + opaque(k3LdsTiles + wave * kHalfTileBytes + 4 * lane);
#pragma unroll
            for (int r = 0; r < kRegs; r++) ab[r] = *(const uint32_t*)(dig_in + 256 * r);
        }
        CUFHE_AMD_PHASE3(2)
        // forward transform and products of both halves, no barrier in between: sums[h] has its own region
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const WaveCtx ctx = half_ctx(h);
            double tu[15];
            load_tu(tu, ctx.tu_l);               // in flight while the digits are unpacked
            double x[kRegs];
#pragma unroll
            for (int r = 0; r < kRegs; r++) {
                uint32_t w = ab[r];
                asm volatile("" : "+v"(w));      // re-derive per half: keeps 16 registers live, not 64
                const double a = (double)(int)(int16_t)(w & 0xffffu), b = (double)((int)w >> 16);
                x[r] = __builtin_fma(h ? -b : b, fpf::ROOT4, a);      // first forward stage, exact: |I b| < 2^33
            }
            // half 0: stage 0 multiplies by psi^512 = zeta = 5440, exact on |x| < 2^33 (2 FMAs instead of a modular product)
            if (h == 0) ntt_forward_a_tu<false, true>(x, tu);
            else ntt_forward_a_tu<false>(x, tu);
            ntt_forward_bc<true, true>(x, ctx);
#pragma unroll
            for (int r = 0; r < kRegs; r++) x[r] = fpf::reduce(x[r]);
            CUFHE_AMD_PHASE3(3 + 2 * h)
#pragma unroll
            for (int pp = 0; pp < k2Prods; pp++) {
                const int k = k2Prods * h + pp;            // compile-time: both loops are unrolled
                if (k + 2 < 2 * k2Prods) load_key_poly(kb[(k + 2) % 3], CUFHE_AMD_KEYPOLY3(key, k + 2), lane);
                accumulate_poly(sumL + h * k3SumDoubles + pp * k2Half, x, kb[k % 3], lane);
            }
            CUFHE_AMD_PHASE3(4 + 2 * h)
        }
        // the twelve inverse jobs in one phase: waves 0-3 -- served first by their SIMDs -- take two sums of one half each,
        // stage by stage in one instruction stream, waves 4-7 one sum each: three jobs per SIMD
        //   wave 0: (3, 4 | h0)   wave 1: (5, 0 | h0)   wave 2: (3, 4 | h1)   wave 3: (5, 0 | h1)
        //   wave 4: (1 | h0)      wave 5: (2 | h0)      wave 6: (1 | h1)      wave 7: (2 | h1)
        const int jh = (wave >> 1) & 1;
        __syncthreads();                                      // the twelve sums are complete
        CUFHE_AMD_PHASE3(7)
        // L2 warming: the 32 workgroups that share an XCD (blocks b, b + 8, ...: round-robin dispatch, for speed only) walk
        // the key together, and whoever touches a line first waits for the fabric.  Each workgroup touches 1/32 of the key of
        // step i + 2 (one 4-byte load per 128-byte line: 768 KiB / 32 = 192 lines = 3 wave-loads on wave 7), so that the
        // XCD's L2 holds the whole step before anyone needs it.  The loads are issued ahead of the inverse jobs and end
        // behind them: nothing waits for them.
        uint32_t warm0 = 0, warm1 = 0, warm2 = 0;
        if (wave == 7 && i + 2 < steps) {
            const char* nk = (const char*)(bk_ntt + (size_t)(i + 2) * k2BkStepDoubles) + (size_t)((blockIdx.x >> 3) & 31) * 24576 + lane * 128;
            warm0 = *(const uint32_t*)nk;
            warm1 = *(const uint32_t*)(nk + 8192);
            warm2 = *(const uint32_t*)(nk + 16384);
        }
        if (wave < 4) inverse_job2((wave & 1) ? 5 : 3, (wave & 1) ? 0 : 4, jh);      // one copy of the code: o, h are run-time values
        else inverse_job(1 + (wave & 1), jh);
        // the key of the next step is requested now (idle vector-memory path; behind the jobs, whose loads would queue behind it)
        asm volatile("" :: "v"(warm0), "v"(warm1), "v"(warm2));      // the warming loads end here, long after their issue
        if (i + 1 < steps) prefetch_key(i + 1);
        CUFHE_AMD_PHASE3(8)
        __syncthreads();
        CUFHE_AMD_PHASE3(9)
        // every wave recombines its pairs of its output into its accumulator registers and leaves a copy for the next rotated read
        recombine();
        publish_acc();
        CUFHE_AMD_PHASE3(12)
    }
#undef CUFHE_AMD_KEYPOLY3


    if (acc_dump) {
        uint64_t* o = acc_dump + (size_t)g * 2 * k2N + wj * k2N + e_first;
#pragma unroll
        for (int m = 0; m < 4; m++) { o[64 * m] = acc_lo[m]; o[64 * m + k2Half] = acc_hi[m]; }
    }
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_PHASES)
    if (acc_dump && lane == 0 && g == 0) {
        __syncthreads();
        unsigned long long* o = (unsigned long long*)acc_dump + 2048 + wave * 16;     // overwrites part of the dump: timing only
        for (int k = 0; k < 16; k++) o[k] = ph[k];
    }
#endif
    if (d.out) {   // __SampleExtractIndex__<lvl2param,0>, src/bootstrap_gpu.cu:366-381
        uint64_t* o = d.out;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int e0 = e_first + 64 * m, e1 = e0 + k2Half;
            if (wj) {
                if (e0 == 0) o[k2N] = acc_lo[m];
            } else {
                if (e0 == 0) o[0] = acc_lo[m];
                else o[k2N - e0] = 0ull - acc_lo[m];
                o[k2N - e1] = 0ull - acc_hi[m];
            }
        }
    }
}

// ----------------------------------------------------------------------------------
// Key switch lvl2 -> lvl0 (KeySwitchFromTLWE<lvl20>, include/keyswitch_gpu.cuh:83-134) with
// the linear pre-add of the Mux fused.  One workgroup (16 waves) per ciphertext, wave w takes
// a'_j for j in [128 w, 128 w + 128); rows are read from L2, the 16 partial sums are added
// through LDS.  Table layout as lvl1: [j][k][v][640] padded rows.
// ----------------------------------------------------------------------------------
constexpr int k2KsStepRows = k2KsT * k2KsNumBase;             // 14 rows per j

__global__ __launch_bounds__(kKsThreads) void keyswitch_lvl2_kernel(
    const LinDesc64* __restrict__ descs, int count, const uint32_t* __restrict__ ksk_padded)
{
    __shared__ uint32_t part[kKsWaves][kKsRowPad];
    __shared__ uint16_t dig[k2N];
    __shared__ uint32_t bprime_s;
    const int g = blockIdx.x;
    if (g >= count) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const LinDesc64 d = descs[g];
    // iksoffsetgen<lvl20> + roundoffset (:13-23,92-98); only the top t*basebit = 14 bits carry digits
    uint64_t koff = 1ull << (64 - (1 + k2KsBasebit * k2KsT));
    for (int i = 1; i <= k2KsT; i++) koff += ((1ull << k2KsBasebit) / 2) << (64 - i * k2KsBasebit);
    for (int j = tid; j < k2Words; j += kKsThreads) {
        const uint64_t v = (uint64_t)(int64_t)d.ca * d.in0[j] + (uint64_t)(int64_t)d.cb * d.in1[j];
        if (j == k2N) bprime_s = (uint32_t)((v + d.off + (1ull << 31)) >> 32);     // rounding narrowing, :100-101
        else dig[j] = (uint16_t)((v + koff) >> 48);
    }
    __syncthreads();

    int piece[kKsPieces];
    piece[0] = lane; piece[1] = lane + 64; piece[2] = lane < 32 ? lane + 128 : 159;
    uint4 res[kKsPieces];
#pragma unroll
    for (int m = 0; m < kKsPieces; m++) res[m] = make_uint4(0, 0, 0, 0);
    const uint4* base = (const uint4*)ksk_padded;
    constexpr int kRowPieces = kKsRowPad / 4;
#pragma unroll 1
    for (int jj = 0; jj < k2N / kKsWaves; jj++) {
        const int j = wave * (k2N / kKsWaves) + jj;
        const uint32_t dj = __builtin_amdgcn_readfirstlane((uint32_t)dig[j]);
        int val[k2KsT];
        uint4 row[k2KsT][kKsPieces];
#pragma unroll
        for (int k = 0; k < k2KsT; k++) {
            val[k] = (int)((dj >> (16 - (k + 1) * k2KsBasebit)) & ((1u << k2KsBasebit) - 1)) - (1 << (k2KsBasebit - 1));
            const int v = val[k] > 0 ? val[k] : -val[k];
            const uint4* r = base + ((size_t)(j * k2KsT + k) * k2KsNumBase + (v ? v - 1 : 0)) * kRowPieces;
#pragma unroll
            for (int m = 0; m < kKsPieces; m++) row[k][m] = r[piece[m]];
        }
#pragma unroll
        for (int k = 0; k < k2KsT; k++) {
            if (val[k] > 0) {
#pragma unroll
                for (int m = 0; m < kKsPieces; m++) { res[m].x -= row[k][m].x; res[m].y -= row[k][m].y; res[m].z -= row[k][m].z; res[m].w -= row[k][m].w; }
            } else if (val[k] < 0) {
#pragma unroll
                for (int m = 0; m < kKsPieces; m++) { res[m].x += row[k][m].x; res[m].y += row[k][m].y; res[m].z += row[k][m].z; res[m].w += row[k][m].w; }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < kKsPieces; m++) {
        if (m == 2 && lane >= 32) break;
        *(uint4*)&part[wave][4 * piece[m]] = res[m];
    }
    __syncthreads();
    for (int i = tid; i <= kLvl0N; i += kKsThreads) {
        uint32_t v = (i == kLvl0N) ? bprime_s : 0u;
#pragma unroll
        for (int w = 0; w < kKsWaves; w++) v += part[w][i];
        d.out[i] = v;
    }
}

}  // namespace cufhe_amd
