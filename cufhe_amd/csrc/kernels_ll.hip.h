// kernels_ll.hip.h -- low-latency blind rotate: one 16-wave workgroup per rotation, every
// transform split into its two 512-point halves (ntt_wave512.h).
//
// Same words as blind_rotate_kernel (kernels.hip.h); used for launches
// too small to fill the chip with the wave-per-rotation kernel, where what counts is the length
// of the dependent chain of ONE rotation: 630 CMux steps, each "decompose -> forward NTT ->
// pointwise -> inverse NTT -> accumulate".
//
//   waves 0-11   (row, h) = (w / 2, w % 2): digit row `row` of the step, half h of its transform:
//                first stage u_h = a[e] +- I a[e + 512] (exact) on the digits the inverse waves left
//                in LDS, 512-point forward transform, 16 products against the two key polynomials of
//                the row, ds_add_f64 into the sums -- three such waves per SIMD
//   waves 12-15  (out, h): inverse 512-point transform of half h of sum `out` -- one per SIMD -- left in LDS
//   all 16 waves the coefficient-wise tail of the step, 128 coefficients (two slices of 64) each: the last
//                inverse stage (u0 + u1, (u0 - u1) I^-1) on the two halves, the centred lift into the
//                accumulator, and -- after a barrier, the rotated operand comes from other waves' slices --
//                the gadget decomposition of the NEXT step, all l digits as signed bytes (each row wave
//                used to recompute the decomposed word of its coefficients: six times the same arithmetic,
//                on the waves that set the length of the step)
// Four workgroup barriers per step; every wave keeps the twiddles of its role in registers for the whole kernel.  The NTT-domain
// key is read in its ordinary layout.
// The kernels of this file are compiled in their own translation unit (kernels_ll.hip, with
// -mllvm -amdgpu-sched-strategy=max-ilp: 3.5 % faster here, while the same strategy costs the N = 512 parameter-set
// kernel 4 %); capi.hip includes it with CUFHE_AMD_LL_DECLARATIONS_ONLY for the constants and the prototypes.
#pragma once
#include "kernels_common.hip.h"
#include "ntt_wave512.h"

namespace cufhe_amd {

constexpr int kLlThreads = 1024;
constexpr int kLlRowWaves = 2 * kBkRows;                                      // 12
constexpr int kLlLdsTables = 0;                                               // [h][tb_fwd|tb_inv|tc_fwd|tc_inv]
constexpr int kLlLdsTiles = kLlLdsTables + 2 * kLds512TableBytes;             // 16128
constexpr int kLlLdsAcc = kLlLdsTiles + 16 * kTile512Bytes;                   // + 72704
constexpr int kLlLdsSum = kLlLdsAcc + 2 * 2 * kN * 4;                         // + 16384   [j][copy][N] u32
constexpr int kLlLdsHand = kLlLdsSum + 2 * kN * 8;                            // + 16384   [out][h][c][lane] f64
constexpr int kLlLdsDig = kLlLdsHand + 2 * kN * 8;                            // + 16384   [out][h][e] f64
constexpr int kLlDigBytes = 2 * kN * 4;                                       // the decomposed words of one rotation, [m][e] u32
constexpr int kLlLdsAbar = kLlLdsDig + kLlDigBytes;                           // + 8192
constexpr int kLlLdsBytes = kLlLdsAbar + kAbarBytes + 16;                     // 147472

__global__ __launch_bounds__(kLlThreads) void blind_rotate_ll_kernel(
    const LinDesc* __restrict__ descs, int count, const double* __restrict__ bk_ntt,
    const Ntt512Tables* __restrict__ gt2, int steps, uint32_t* __restrict__ acc_dump);
#ifndef CUFHE_AMD_LL_DECLARATIONS_ONLY
// The first TWO stages of half h of the forward transform on one row's gadget digits, exactly, from the DECOMPOSED WORDS of the
// coefficients: w0[r] / w1[r] hold the word of e = lane + 64 r and e + 512 with the sign mask applied, the row's digit is the signed field
// at bit `pos` (include/gatebootstrapping_gpu.cuh:157-181).  The 1024-point transform's stage 0 gives u_h = a +- I b, its stage 1 pairs
// u_h[e] with u_h[e + 256] under zeta (h = 0) or zeta^3 (h = 1).  On the four original digits a, a' (e, e + 256) and b, b' (e + 512,
// e + 768) both are exact linear forms -- zeta^3 I = zeta^5 = -zeta, so zeta^3 (a' - I b') = zeta^3 a' + zeta b' and the 37-bit root only
// meets a 6-bit digit: five FMAs per pair where the general butterfly took ten operations.  |x| < 2^42.2.  (The tail stores one dword per
// coefficient and component -- lane-contiguous, conflict-free -- instead of l digit bytes into words that eight lanes share a bank for,
// and nothing has to be packed.)
__device__ __forceinline__ void ll_split_first_stages_words(double (&x)[kRegs8], const uint32_t (&w0)[8], const uint32_t (&w1)[8], int h, uint32_t pos)
{
    constexpr double kZ3 = fpf::ROOT8 * fpf::ROOT8 * fpf::ROOT8;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double a = (double)(int32_t)__builtin_amdgcn_sbfe(w0[r], pos, (uint32_t)kBgbit), a1 = (double)(int32_t)__builtin_amdgcn_sbfe(w0[r + 4], pos, (uint32_t)kBgbit);
        const double b = (double)(int32_t)__builtin_amdgcn_sbfe(w1[r], pos, (uint32_t)kBgbit), b1 = (double)(int32_t)__builtin_amdgcn_sbfe(w1[r + 4], pos, (uint32_t)kBgbit);
        if (h == 0) {
            const double u = __builtin_fma(b, fpf::ROOT4, a), u1 = __builtin_fma(b1, fpf::ROOT4, a1);
            x[r] = __builtin_fma(u1, fpf::ROOT8, u);
            x[r + 4] = __builtin_fma(-u1, fpf::ROOT8, u);
        } else {
            const double v = __builtin_fma(-b, fpf::ROOT4, a);
            const double t = __builtin_fma(a1, kZ3, b1 * fpf::ROOT8);
            x[r] = v + t;
            x[r + 4] = v - t;
        }
    }
}

__global__ __launch_bounds__(kLlThreads) void blind_rotate_ll_kernel(
    const LinDesc* __restrict__ descs, int count, const double* __restrict__ bk_ntt,
    const Ntt512Tables* __restrict__ gt2, int steps, uint32_t* __restrict__ acc_dump)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = blockIdx.x;
    if (g >= count) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    uint32_t* accL = (uint32_t*)(smem + kLlLdsAcc);           // [j][copy][N]
    double* sumL = (double*)(smem + kLlLdsSum);               // [out][h][c][lane]
    double* handL = (double*)(smem + kLlLdsHand);             // [out][h][e]
    uint16_t* abar_lds = (uint16_t*)(smem + kLlLdsAbar);
    uint32_t* bbar_slot = (uint32_t*)(smem + kLlLdsAbar + kAbarBytes);

    for (int i = tid; i < 2 * kLds512TableDoubles; i += kLlThreads) {     // tb_fwd .. tc_inv are contiguous
        const int h = i / kLds512TableDoubles, k = i % kLds512TableDoubles;
        ((double*)(smem + kLlLdsTables))[i] = gt2[h].tb_fwd[k];
    }
    const LinDesc d = descs[g];
    for (int i = tid; i <= kLvl0N; i += kLlThreads) {
        const uint32_t c = (uint32_t)d.ca * d.in0[i] + (uint32_t)d.cb * d.in1[i];
        if (i < kLvl0N) abar_lds[i] = (uint16_t)((c + (1u << (32 - 2 - kNbit))) >> (32 - 1 - kNbit));
        else *bbar_slot = 2 * kN - ((c + d.off) >> (32 - 1 - kNbit));
    }
    for (int i = tid; i < 2 * kN; i += kLlThreads) sumL[i] = 0.0;
    __syncthreads();
    {   // RotatedTestVector, include/gatebootstrapping_gpu.cuh:29-52
        const uint32_t bbar = *bbar_slot;
        for (int e = tid; e < kN; e += kLlThreads) {
            const bool neg = (bbar != 2 * kN) && (((uint32_t)e < (bbar & (kN - 1))) != ((bbar >> kNbit) != 0));
            const uint32_t v = neg ? 0u - kMu : kMu;
            accL[e] = 0; accL[kN + e] = 0;
            accL[2 * kN + e] = v; accL[3 * kN + e] = v;
        }
    }
    __syncthreads();

    uint32_t* digW = (uint32_t*)(smem + kLlLdsDig);           // [m][hh][rr][lane]: the decomposed word of coefficient lane + 64 rr + 512 hh of component m
    const bool row_wave = wave < kLlRowWaves;
    const int h = wave & 1;                                   // half transform of this wave (both roles)
    const int row = wave >> 1;                                // row waves: TRGSW row = wj * l + wd
    const int wj = row / kL, wd = row % kL;
    const int out = (wave - kLlRowWaves) >> 1;                // inverse waves
    const Wave512Ctx ctx = make_wave512_ctx(smem, kLlLdsTiles + wave * kTile512Bytes,
                                            kLlLdsTables + h * kLds512TableBytes, gt2 + h, lane);
    // Spectrum position p = 512 h + 64 lam + 8 kap + c of the full transform sits in the key
    // layout [q = p[3:1]][lane = p[9:6] | p[5:4] << 4][p[0]]: this lane (lam = lane & 7,
    // kap = lane >> 3) reads, for its registers c = 2 cc, 2 cc + 1, the double2 at
    // q = 4 (kap & 1) + cc, key lane = 8 h + lam + 16 (kap >> 1).
    const int key_idx = (4 * ((lane >> 3) & 1)) * 64 + (8 * h + (lane & 7) + 16 * (lane >> 4));
    double2 b[8];                                             // [out][cc]
    auto load_row = [&](int step) {
        const double2* rowp = (const double2*)(bk_ntt + ((size_t)step * kBkRows + row) * (2 * kN));
#pragma unroll
        for (int o = 0; o < 2; o++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) b[4 * o + cc] = rowp[o * (kN / 2) + key_idx + 64 * cc];
    };
    if (row_wave && steps > 0) load_row(0);
    double tu[7];                                             // this wave's stage 0-2 twiddles: forward for a row wave, inverse otherwise
#pragma unroll
    for (int k = 0; k < 7; k++) tu[k] = row_wave ? gt2[h].tu_fwd[k] : gt2[h].tu_inv[k];
    double twb[7], twc[7];                                    // ... and its per-lam / per-lane twiddles, fetched once (ntt_wave512.h)
#pragma unroll
    for (int k = 0; k < 7; k++) {
        twb[k] = lds_ld(row_wave ? ctx.tb_fwd : ctx.tb_inv, 64 * k);
        twc[k] = lds_ld(row_wave ? ctx.tc_fwd : ctx.tc_inv, 512 * k);
    }
    if (!row_wave) {      // the inverse transform runs in radix-4 form: slot 2 of each block holds the product v w (no register more)
        tu[2] = gt2[h].tu_inv[7];
        twb[2] = gt2[h].uwb_inv[lane & 7];
        twc[2] = gt2[h].uwc_inv[lane];
    }
    // The coefficient-wise tail of a step is spread over all 16 waves: wave k owns the slices (out, hh, rr) =
    // (m, k >> 3, k & 7), m = 0, 1, i.e. coefficients lane + 64 rr + 512 hh of accumulator component m.
    const int hh = wave >> 3, rr = wave & 7;
    const int ecoef = lane + 64 * rr + kH * hh;               // this lane's coefficient, both components
    // digits of (X^abar - 1) acc_m at ecoef for the CMux step `step`, w[m] = acc_m[ecoef]
    // (include/gatebootstrapping_gpu.cuh:157-181); byte (row, hh, lane, rr) of the digit table
    auto decompose = [&](uint32_t abar, const uint32_t (&w)[2]) {
        const int alo = (int)(abar & (kN - 1));
        const bool neg = (ecoef < alo) != ((abar >> kNbit) != 0);
        const int ridx = (ecoef - alo) & (kN - 1);
        uint32_t rot[2];
#pragma unroll
        for (int m = 0; m < 2; m++) rot[m] = accL[m * 2 * kN + ridx];
        uint32_t* dst = digW + (hh * 8 + rr) * 64 + lane;
#pragma unroll
        for (int m = 0; m < 2; m++)
            dst[m * (2 * 8 * 64)] = ((neg ? 0u - rot[m] : rot[m]) - w[m] + decomp_offset()) ^ decomp_signmask();
    };
    if (steps > 0) {
        uint32_t w[2];
#pragma unroll
        for (int m = 0; m < 2; m++) w[m] = accL[m * 2 * kN + ecoef];
        decompose(__builtin_amdgcn_readfirstlane((uint32_t)abar_lds[0]), w);
    }
    // abar of the coming steps in one VGPR per wave, 64 steps to a window, read with v_readlane: no LDS round trip (and no
    // readfirstlane behind it) between the barrier after the accumulator update and the rotated read of the decomposition
    uint32_t abar_win = abar_lds[lane];
    __syncthreads();

#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_PHASES)
    // timing-only: cycles this wave spends in each phase of a step (phase = work up to the next barrier, then the barrier)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#define CUFHE_AMD_PHASE(k) { const unsigned long long tn = __builtin_readcyclecounter(); ph[k] += tn - tc; tc = tn; }
#else
#define CUFHE_AMD_PHASE(k)
#endif
#pragma unroll 1
    for (int i = 0; i < steps; i++) {
        if (((i + 1) & 63) == 0) abar_win = abar_lds[i + 1 + lane];       // entries 630..639 are padding, never used
        const uint32_t abar_next = __builtin_amdgcn_readlane(abar_win, (i + 1) & 63);
        if (row_wave) {
            // the decomposed words of component wj at e = lane + 64 r and e + 512, left by the tail; this row's digit is the
            // field at bit 32 - (wd + 1) Bgbit; then the first two stages of the transform, exactly
            const uint32_t* dgp = digW + wj * (2 * 8 * 64) + lane;
            uint32_t w0[8], w1[8];
#pragma unroll
            for (int r = 0; r < 8; r++) { w0[r] = dgp[r * 64]; w1[r] = dgp[(8 + r) * 64]; }
            double x[kRegs8];
            ll_split_first_stages_words(x, w0, w1, h, 32u - (uint32_t)(wd + 1) * kBgbit);
            ntt512_forward_pinned_from1(x, ctx, tu, twb, twc);
            double* s0 = sumL + h * kH + lane;
#pragma unroll
            for (int o = 0; o < 2; o++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
                    __hip_atomic_fetch_add(s0 + o * kN + (2 * cc) * 64, fpf::mulmod_wide(x[2 * cc], b[4 * o + cc].x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(s0 + o * kN + (2 * cc + 1) * 64, fpf::mulmod_wide(x[2 * cc + 1], b[4 * o + cc].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
        }
        CUFHE_AMD_PHASE(0)
        __syncthreads();
        CUFHE_AMD_PHASE(1)
        double u[kRegs8];
        if (row_wave) {
            if (i + 1 < steps) load_row(i + 1);               // in flight during the inverse transforms
        } else {
            double* s = sumL + out * kN + h * kH + lane;
#pragma unroll
            for (int r = 0; r < kRegs8; r++) { u[r] = fpf::reduce(s[r * 64]); s[r * 64] = 0.0; }
            ntt512_inverse_r4_pinned(u, ctx, tu, twb, twc);      // u_h[e], e = lane + 64 r, |u| <= p
            double* hd = handL + out * kN + h * kH + lane;
#pragma unroll
            for (int r = 0; r < kRegs8; r++) hd[r * 64] = u[r];
        }
        CUFHE_AMD_PHASE(2)
        __syncthreads();
        CUFHE_AMD_PHASE(3)
        uint32_t wnew[2];
        {
            // last inverse stage (u0, u1) -> (u0 + u1, (u0 - u1) I^-1), I^-1 = -I, for coefficient lane + 64 rr of half hh
            // of both components, then the centred lift into the accumulator (both copies)
            const double* hd = handL + rr * 64 + lane;
            double u0[2], u1[2];
            uint32_t old[2];
#pragma unroll
            for (int m = 0; m < 2; m++) {
                u0[m] = hd[m * kN];
                u1[m] = hd[m * kN + kH];
                old[m] = accL[m * 2 * kN + ecoef];
            }
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const double y = hh ? fpf::mulmod(u0[m] - u1[m], -fpf::ROOT4) : u0[m] + u1[m];
                wnew[m] = old[m] + fpf::lift_u32_small(y);
                accL[m * 2 * kN + ecoef] = wnew[m];
                accL[m * 2 * kN + kN + ecoef] = wnew[m];
            }
        }
        CUFHE_AMD_PHASE(4)
        __syncthreads();
        CUFHE_AMD_PHASE(5)
        if (i + 1 < steps) decompose(abar_next, wnew);        // reads other waves' new words: after the barrier
        CUFHE_AMD_PHASE(6)
        __syncthreads();
        CUFHE_AMD_PHASE(7)
    }

    if (acc_dump) {
        uint32_t* o = acc_dump + (size_t)g * 2 * kN;
        for (int e = tid; e < kN; e += kLlThreads) { o[e] = accL[e]; o[kN + e] = accL[2 * kN + e]; }
    }
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_PHASES)
    if (acc_dump && lane == 0 && g == 0) {
        __syncthreads();
        unsigned long long* o = (unsigned long long*)acc_dump + 16 + wave * 8;      // overwrites part of the dump: timing only
        for (int k = 0; k < 8; k++) o[k] = ph[k];
    }
#endif
    if (d.out) {
        uint32_t* o = d.out;      // __SampleExtractIndex__<P,0>
        for (int e = tid; e < kN; e += kLlThreads) {
            if (e == 0) { o[0] = accL[0]; o[kN] = accL[2 * kN]; }
            else o[kN - e] = 0u - accL[e];
        }
    }
}
#endif

// ----------------------------------------------------------------------------------
// Two rotations per workgroup (launches of 257 .. 1536 rotations and their tails): the same waves, the same
// arithmetic, but while the twelve row waves work on one rotation the four inverse waves transform the sums of the
// other, and the two rotations share every key row (loaded once into the row waves' registers).
//   slot 1   row waves: row phase of A, step i     inverse waves: tail of B's step i-1 (inverse transforms, last stage,
//                                                   lift, accumulator, decomposition for step i)
//   slot 2   row waves: row phase of B, step i     inverse waves: the same for A's step i
// Two workgroup barriers per step for two rotations; inside a tail the four inverse waves meet twice at a counter in
// LDS.  LDS per rotation: accumulator 8 KiB (one copy: the decomposition computes its rotated index), sums 16 KiB (an inverse wave leaves its half transform where it read its sum), digits
// 6 KiB, abar list.
// ----------------------------------------------------------------------------------
constexpr int kLl2RotBytes = 2 * kN * 4 + 2 * kN * 8 + kLlDigBytes + kAbarBytes + 16;            // 34080
constexpr int kLl2LdsRot = kLlLdsTiles + 16 * kTile512Bytes;
constexpr int kLl2LdsBytes = kLl2LdsRot + 2 * kLl2RotBytes + 16;                                  // 157008 (+ the inverse waves' counter)
static_assert(kLl2LdsBytes <= 160 * 1024, "paired low-latency kernel does not fit the CU's LDS");

__global__ __launch_bounds__(kLlThreads) void blind_rotate_ll2_kernel(
    const LinDesc* __restrict__ descs, int count, const double* __restrict__ bk_ntt,
    const Ntt512Tables* __restrict__ gt2, int steps, uint32_t* __restrict__ acc_dump, uint32_t* fault);
#ifndef CUFHE_AMD_LL_DECLARATIONS_ONLY
__global__ __launch_bounds__(kLlThreads) void blind_rotate_ll2_kernel(
    const LinDesc* __restrict__ descs, int count, const double* __restrict__ bk_ntt,
    const Ntt512Tables* __restrict__ gt2, int steps, uint32_t* __restrict__ acc_dump, uint32_t* fault)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    struct Rot { uint32_t* acc; double* sum; uint32_t* dig; uint16_t* abar; uint32_t* bbar; };
    Rot rot[2];
    int gidx[2];
#pragma unroll
    for (int x = 0; x < 2; x++) {
        char* base = smem + kLl2LdsRot + x * kLl2RotBytes;
        rot[x].acc = (uint32_t*)base;                                        // [j][N]
        rot[x].sum = (double*)(base + 2 * kN * 4);                           // [out][h][c][lane], then [out][h][e]
        rot[x].dig = (uint32_t*)(base + 2 * kN * 4 + 2 * kN * 8);            // [m][e]: the decomposed word of coefficient e of component m
        rot[x].abar = (uint16_t*)(base + 2 * kN * 4 + 2 * kN * 8 + kLlDigBytes);
        rot[x].bbar = (uint32_t*)((char*)rot[x].abar + kAbarBytes);
        const int g = 2 * (int)blockIdx.x + x;
        gidx[x] = g < count ? g : count - 1;             // an odd launch computes its last rotation twice (second copy: no output)
    }
    if (2 * (int)blockIdx.x >= count) return;
    const bool live1 = 2 * (int)blockIdx.x + 1 < count;
    if (tid == 0) *(uint32_t*)(smem + kLl2LdsRot + 2 * kLl2RotBytes) = 0;      // the inverse waves' counter (inv_sync)

    for (int i = tid; i < 2 * kLds512TableDoubles; i += kLlThreads) {     // tb_fwd .. tc_inv are contiguous
        const int hh = i / kLds512TableDoubles, k = i % kLds512TableDoubles;
        ((double*)(smem + kLlLdsTables))[i] = gt2[hh].tb_fwd[k];
    }
    LinDesc dsc[2];
#pragma unroll
    for (int x = 0; x < 2; x++) {
        dsc[x] = descs[gidx[x]];
        for (int i = tid; i <= kLvl0N; i += kLlThreads) {
            const uint32_t c = (uint32_t)dsc[x].ca * dsc[x].in0[i] + (uint32_t)dsc[x].cb * dsc[x].in1[i];
            if (i < kLvl0N) rot[x].abar[i] = (uint16_t)((c + (1u << (32 - 2 - kNbit))) >> (32 - 1 - kNbit));
            else *rot[x].bbar = 2 * kN - ((c + dsc[x].off) >> (32 - 1 - kNbit));
        }
        for (int i = tid; i < 2 * kN; i += kLlThreads) rot[x].sum[i] = 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 2; x++) {   // RotatedTestVector, include/gatebootstrapping_gpu.cuh:29-52
        const uint32_t bbar = *rot[x].bbar;
        for (int e = tid; e < kN; e += kLlThreads) {
            const bool neg = (bbar != 2 * kN) && (((uint32_t)e < (bbar & (kN - 1))) != ((bbar >> kNbit) != 0));
            rot[x].acc[e] = 0;
            rot[x].acc[kN + e] = neg ? 0u - kMu : kMu;
        }
    }
    __syncthreads();

    const bool row_wave = wave < kLlRowWaves;
    const int h = wave & 1;
    const int row = wave >> 1;
    const int out = (wave - kLlRowWaves) >> 1;
    const Wave512Ctx ctx = make_wave512_ctx(smem, kLlLdsTiles + wave * kTile512Bytes,
                                            kLlLdsTables + h * kLds512TableBytes, gt2 + h, lane);
    const int key_idx = (4 * ((lane >> 3) & 1)) * 64 + (8 * h + (lane & 7) + 16 * (lane >> 4));
    double2 b[8];
    auto load_row = [&](int step) {
        const double2* rowp = (const double2*)(bk_ntt + ((size_t)step * kBkRows + row) * (2 * kN));
#pragma unroll
        for (int o = 0; o < 2; o++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) b[4 * o + cc] = rowp[o * (kN / 2) + key_idx + 64 * cc];
    };
    if (row_wave && steps > 0) load_row(0);
    double tu[7];                                             // this wave's stage 0-2 twiddles: forward for a row wave, inverse otherwise
#pragma unroll
    for (int k = 0; k < 7; k++) tu[k] = row_wave ? gt2[h].tu_fwd[k] : gt2[h].tu_inv[k];
    double twb[7], twc[7];                                    // ... and its per-lam / per-lane twiddles, fetched once
#pragma unroll
    for (int k = 0; k < 7; k++) {
        twb[k] = lds_ld(row_wave ? ctx.tb_fwd : ctx.tb_inv, 64 * k);
        twc[k] = lds_ld(row_wave ? ctx.tc_fwd : ctx.tc_inv, 512 * k);
    }
    if (!row_wave) {      // the inverse transform runs in radix-4 form: slot 2 of each block holds the product v w (no register more)
        tu[2] = gt2[h].tu_inv[7];
        twb[2] = gt2[h].uwb_inv[lane & 7];
        twc[2] = gt2[h].uwc_inv[lane];
    }
    const int hh = wave >> 3, rr = wave & 7;
    const int ecoef = lane + 64 * rr + kH * hh;

    // row waves: digits -> first stage -> forward half transform -> products into the sums of rotation r
    auto row_phase = [&](const Rot& r) {
        const uint32_t* dgp = r.dig + (row / kL) * kN + lane;
        uint32_t w0[8], w1[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { w0[k] = dgp[k * 64]; w1[k] = dgp[kH + k * 64]; }
        double x[kRegs8];
        ll_split_first_stages_words(x, w0, w1, h, 32u - (uint32_t)(row % kL + 1) * kBgbit);
        ntt512_forward_pinned_from1(x, ctx, tu, twb, twc);
        double* s0 = r.sum + h * kH + lane;
#pragma unroll
        for (int o = 0; o < 2; o++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                __hip_atomic_fetch_add(s0 + o * kN + (2 * cc) * 64, fpf::mulmod_wide(x[2 * cc], b[4 * o + cc].x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(s0 + o * kN + (2 * cc + 1) * 64, fpf::mulmod_wide(x[2 * cc + 1], b[4 * o + cc].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
    };
    // The four inverse waves synchronise among themselves through a counter in LDS (the row waves are busy with the
    // other rotation and must not be held up by a workgroup barrier -- measured: with the tail of a step spread over all
    // sixteen waves behind two more barriers a round of 512 rotations takes 6 % longer): every wave adds 1, then waits
    // for 4 more than last time.  DS operations of a wave execute in order, so what it wrote before the add is in LDS
    // when the count shows.  A correct run waits a few hundred cycles (the waves of a workgroup are resident, and
    // preempted, together).  The wait is bounded (2^20 polls with s_sleep, about 0.1 s) so that a logic error cannot hang
    // the device -- and a wave whose wait expires REPORTS it: it sets kFaultLl2SyncTimeout in the device's fault word
    // (host-visible memory; Synchronize / StreamQuery / every completion the scheduler observes return status -5 from
    // then on, the C++ shim aborts) and poisons the counter, so that the workgroup drains at once instead of timing out
    // 2 x 630 more times.  The words such a launch produced are wrong; nothing downstream may trust them, and with the
    // status nothing has to find that out from a failed decryption.
    uint32_t* sync_cnt = (uint32_t*)(smem + kLl2LdsRot + 2 * kLl2RotBytes);
    uint32_t sync_target = 0;
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_LL2_TIMEOUT)
    constexpr int kSyncSpins = 1 << 12;      // fault injection (tests/test_gpu_fault.py): wave 15 of workgroup 0 skips one arrival
    int sync_calls = 0;
#else
    constexpr int kSyncSpins = 1 << 20;
#endif
    auto inv_sync = [&]() {
        sync_target += 4;
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_LL2_TIMEOUT)
        const bool skip = blockIdx.x == 0 && wave == 15 && ++sync_calls == 5;
        if (lane == 0 && !skip)
#else
        if (lane == 0)
#endif
            __hip_atomic_fetch_add(sync_cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        int spins = 0;
        for (; spins < kSyncSpins; spins++) {
            if (__hip_atomic_load(sync_cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= sync_target) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (spins == kSyncSpins && lane == 0) {
            __hip_atomic_fetch_or(sync_cnt, 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);     // every later wait passes
            if (fault) __hip_atomic_fetch_or(fault, kFaultLl2SyncTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    // inverse waves: the whole tail of rotation r's step -- half h of sum `out` (left in place of the sum for the
    // sibling half), last stage + lift + accumulator at coefficients lane + 64 k + 512 h, and, once all four have stored
    // their words, the decomposition of those coefficients for step `next_step` (all l digits, eight signed bytes to a word)
    auto inverse_chain = [&](const Rot& r, int next_step, bool more) {
        double* s = r.sum + out * kN + h * kH + lane;
        double u[kRegs8];
#pragma unroll
        for (int k = 0; k < kRegs8; k++) u[k] = fpf::reduce(s[k * 64]);
        ntt512_inverse_r4_pinned(u, ctx, tu, twb, twc);
#pragma unroll
        for (int k = 0; k < kRegs8; k++) s[k * 64] = u[k];
        inv_sync();
        const double* other = r.sum + out * kN + (h ^ 1) * kH + lane;
        uint32_t* acck = r.acc + out * kN + h * kH + lane;
        uint32_t wnew[kRegs8];
#pragma unroll
        for (int k = 0; k < kRegs8; k++) {
            const double v = other[k * 64];
            const double y = h ? fpf::mulmod(v - u[k], -fpf::ROOT4) : u[k] + v;
            wnew[k] = acck[k * 64] + fpf::lift_u32_small(y);
            acck[k * 64] = wnew[k];
        }
        inv_sync();
#pragma unroll
        for (int k = 0; k < kRegs8; k++) s[k * 64] = 0.0;       // the sibling has read it
        if (!more) return;
        const uint32_t abar = __builtin_amdgcn_readfirstlane((uint32_t)r.abar[next_step]);
        const int alo = (int)(abar & (kN - 1));
        const bool ahi = (abar >> kNbit) != 0;
        const uint32_t* accj = r.acc + out * kN;
        uint32_t rv[kRegs8];
#pragma unroll
        for (int k = 0; k < kRegs8; k++) rv[k] = accj[(lane + 64 * k + h * kH - alo) & (kN - 1)];
        // the decomposed word itself, one lane-contiguous dword per coefficient: the row waves take their digit field out of it
        uint32_t* dgo = r.dig + out * kN + h * kH + lane;
#pragma unroll
        for (int k = 0; k < kRegs8; k++) {
            const bool neg = (lane + 64 * k + h * kH < alo) != ahi;
            dgo[64 * k] = ((neg ? 0u - rv[k] : rv[k]) - wnew[k] + decomp_offset()) ^ decomp_signmask();
        }
    };
    // one serial chain per slot: the inverse waves go first whenever they can issue (without it the chain, behind
    // three row waves per SIMD, sets the length of the slot: 5.9 ms per 2 rotations against 4.8)
    if (!row_wave) __builtin_amdgcn_s_setprio(3);

    // digits of step 0, both rotations
    if (steps > 0) {
#pragma unroll
        for (int x = 0; x < 2; x++) {
            uint32_t w[2];
#pragma unroll
            for (int m = 0; m < 2; m++) w[m] = rot[x].acc[m * kN + ecoef];
            const uint32_t abar = __builtin_amdgcn_readfirstlane((uint32_t)rot[x].abar[0]);
            const int alo = (int)(abar & (kN - 1));
            const bool neg = (ecoef < alo) != ((abar >> kNbit) != 0);
            const int ridx = (ecoef - alo) & (kN - 1);
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const uint32_t rv = rot[x].acc[m * kN + ridx];
                rot[x].dig[m * kN + ecoef] = ((neg ? 0u - rv : rv) - w[m] + decomp_offset()) ^ decomp_signmask();
            }
        }
    }
    __syncthreads();

    // One loop per role (the role of a wave never changes): the registers of a row wave -- key row, twiddles, spectrum -- and
    // those of an inverse wave are allocated separately instead of as the union of both over one loop.
    //   slot 1: rows of A (step i) beside the tail of B's step i - 1 (inverse transforms, accumulator, digits of step i)
    //   slot 2: rows of B (step i) beside the tail of A's step i
    if (row_wave) {
#pragma unroll 1
        for (int i = 0; i < steps; i++) {
            row_phase(rot[0]);
            __syncthreads();
            row_phase(rot[1]);
            if (i + 1 < steps) load_row(i + 1);               // b is free
            __syncthreads();
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < steps; i++) {
            if (i > 0) inverse_chain(rot[1], i, true);
            __syncthreads();
            inverse_chain(rot[0], i + 1, i + 1 < steps);
            __syncthreads();
        }
    }
    if (steps > 0) {      // B's last step
        if (!row_wave) inverse_chain(rot[1], steps, false);
        __syncthreads();
    }

#pragma unroll
    for (int x = 0; x < 2; x++) {
        if (x == 1 && !live1) break;
        const int g = 2 * (int)blockIdx.x + x;
        if (acc_dump) {
            uint32_t* o = acc_dump + (size_t)g * 2 * kN;
            for (int e = tid; e < 2 * kN; e += kLlThreads) o[e] = rot[x].acc[e];
        }
        if (dsc[x].out) {
            uint32_t* o = dsc[x].out;      // __SampleExtractIndex__<P,0>
            for (int e = tid; e < kN; e += kLlThreads) {
                if (e == 0) { o[0] = rot[x].acc[0]; o[kN] = rot[x].acc[kN]; }
                else o[kN - e] = 0u - rot[x].acc[e];
            }
        }
    }
}
#endif

// ----------------------------------------------------------------------------------
// res = a (signed small) * b (torus) mod (X^512 + 1, 2^32), one wave per product: the
// stand-alone 512-point negacyclic transform (the reference's SmallForwardNTT_512 /
// SmallInverseNTT_512, include/ntt_gpu/ntt_gpuntt.cuh:283-329,394-440) -- the same wave code as
// a half transform, with the table of psi_1024 = psi_2048^2.  Parity hook.
// ----------------------------------------------------------------------------------
constexpr int kPoly512LdsBytes = kLds512TableBytes + kNttWavesPerBlock * kTile512Bytes;

__global__ __launch_bounds__(kNttThreads) void polymul512_kernel(
    uint32_t* __restrict__ res, const int32_t* __restrict__ a, const uint32_t* __restrict__ b,
    int count, const Ntt512Tables* __restrict__ gt, double n_inverse);
#ifndef CUFHE_AMD_LL_DECLARATIONS_ONLY
__global__ __launch_bounds__(kNttThreads) void polymul512_kernel(
    uint32_t* __restrict__ res, const int32_t* __restrict__ a, const uint32_t* __restrict__ b,
    int count, const Ntt512Tables* __restrict__ gt, double n_inverse)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < kLds512TableDoubles; i += kNttThreads) ((double*)smem)[i] = gt->tb_fwd[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * kNttWavesPerBlock + wave;
    if (g >= count) return;
    const Wave512Ctx ctx = make_wave512_ctx(smem, kLds512TableBytes + wave * kTile512Bytes, 0, gt, lane);
    double x[kRegs8], y[kRegs8];
#pragma unroll
    for (int r = 0; r < kRegs8; r++) {
        x[r] = (double)a[(size_t)g * kH + lane + 64 * r];
        y[r] = (double)(int32_t)b[(size_t)g * kH + lane + 64 * r];
    }
    ntt512_forward(x, ctx);
    ntt512_forward(y, ctx);
#pragma unroll
    for (int r = 0; r < kRegs8; r++) {
        y[r] = fpf::reduce(fpf::mulmod_wide(y[r], n_inverse));
        x[r] = fpf::reduce(fpf::mulmod_wide(x[r], y[r]));
    }
    ntt512_inverse(x, ctx);
#pragma unroll
    for (int r = 0; r < kRegs8; r++) res[(size_t)g * kH + lane + 64 * r] = fpf::lift_u32(x[r]);
}
#endif

}  // namespace cufhe_amd
