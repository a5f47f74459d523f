// kernels.hip.h -- gfx950 kernels of the gate-bootstrapping path.
//
//   bk_to_ntt_kernel       __TRGSW2NTT__            src/bootstrap_gpu.cu:43-70
//   blind_rotate_kernel    __BlindRotatePreAdd__ / __BlindRotate__ + Accumulate +
//                          __SampleExtractIndex__   include/gatebootstrapping_gpu.cuh:115-345,
//                                                   src/bootstrap_gpu.cu:366-381
//   keyswitch_kernel       KeySwitchFromTLWE / IdentityKeySwitchPreAdd
//                                                   include/keyswitch_gpu.cuh:83-188
//   lincomb_kernel         __NotBootstrap__/__CopyBootstrap__ and the Mux/NMux sums
//                                                   src/bootstrap_gpu.cu:681-703,728-740
//   keyswitch_wg_kernel / keyswitch_split_kernel   the key switch with one / eight workgroups per ciphertext (small launches;
//                                                  the low-latency blind rotations are in kernels_ll.hip.h)
//                                                   (small launches: lowest latency)
//   sample_extract_kernel, cmux_kernel             SEIandKS / Refresh / CMUXNTT pieces
//                                                   src/keyswitch_gpu.cu:26-40, src/bootstrap_gpu.cu:197-285
//   polymul_kernel         the NTT product check of test/test_polynomial_mult_1024.cu:76-99
//
// Execution model (not the reference's one-block-per-gate/one-launch-per-gate): a launch
// covers a whole batch; ONE WAVEFRONT owns one blind rotation from the first CMux to the
// sample extract.  Its accumulator (2 x 1024 torus words) and the two NTT-domain sums
// (2 x 1024 residues) stay in VGPRs for all n = 630 steps.  The 8 waves of a workgroup walk
// the bootstrapping key together: each 16 KiB TRGSW row is brought into LDS once per
// workgroup by LDS-DMA and read from there by all 8; the row barrier is the only
// synchronisation, the NTTs themselves need none.
#pragma once
#include "kernels_common.hip.h"

namespace cufhe_amd {

// ----------------------------------------------------------------------------------
// BK -> NTT domain.  One wave per torus polynomial.  Values are read as SIGNED words
// (see fpfield.h), transformed, scaled by N^-1 (so the inverse transform needs no
// scaling) and stored centred, in layout C order: [poly][q = reg/2][lane][reg & 1].
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kNttThreads) void bk_to_ntt_kernel(
    double* __restrict__ bk_ntt, const uint32_t* __restrict__ bk, size_t polys,
    const NttTables* __restrict__ gt, double n_inverse)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tabs = (double*)smem;
    load_tables_to_lds(tabs, gt);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t poly = (size_t)blockIdx.x * kNttWavesPerBlock + wave;
    if (poly >= polys) return;
    const WaveCtx ctx = make_wave_ctx(smem, kLdsTableBytes + wave * kTileBytes, 0, gt, lane);
    double x[kRegs];
#pragma unroll
    for (int r = 0; r < kRegs; r++) x[r] = (double)(int32_t)bk[poly * kN + lane + 64 * r];
    ntt_forward<false>(x, ctx);
    double2* dst = (double2*)(bk_ntt + poly * kN);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        double2 v;
        v.x = fpf::reduce(fpf::mulmod_wide(x[2 * q], n_inverse));
        v.y = fpf::reduce(fpf::mulmod_wide(x[2 * q + 1], n_inverse));
        dst[q * 64 + lane] = v;
    }
}

// ----------------------------------------------------------------------------------
// Shader clock under an FP64 load: every wave runs a dependent FMA chain for about a millisecond between two pairs of
// stamps, shader cycles (s_memtime) over the constant 100 MHz counter (s_memrealtime).  A measurement aid for
// cufhe_amd_probe_clock (bench.py prices its instruction counts with THIS run's clock, not a recorded one).
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void clock_probe_kernel(double* __restrict__ hz_out, double* __restrict__ sink, int iters)
{
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) { x = __builtin_fma(x, 0.999999, y); y = __builtin_fma(y, 0.999999, x); }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (x + y == 12345.678) sink[0] = x;       // keeps the chain
    if ((threadIdx.x & 63) == 0)
        hz_out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (double)(c1 - c0) / (double)(r1 - r0) * 1.0e8;
}

// ----------------------------------------------------------------------------------
// Blind rotate + sample extract, one wave per rotation.
// ----------------------------------------------------------------------------------
// acc_j -> (X^abar - 1) acc_j + gadget offset, include/gatebootstrapping_gpu.cuh:157-181.
// The rotation goes through the wave's LDS tile, and so does its sign: the tile receives
// B[0..N) = -s acc, B[N..2N) = +s acc with s = -1 if abar >= N else +1.  Coefficient e of
// X^abar acc is then simply B[e - (abar mod N) + N]: no wrap-around arithmetic, no compare,
// no per-element negation (the reference selects the sign per element, :165-168).
__device__ __forceinline__ void rotate_sub(uint32_t (&temp)[kRegs], const uint32_t (&acc)[kRegs],
                                           char* tile, int lane, uint32_t abar)
{
    const int alo = (int)(abar & (kN - 1));
    const bool ahi = (abar >> kNbit) != 0;
    char* wpos = tile + opaque(4 * lane + (ahi ? 0 : 4 * kN));      // where +acc goes
    char* wneg = tile + opaque(4 * lane + (ahi ? 4 * kN : 0));      // where -acc goes
    asm volatile("" ::: "memory");      // compiler-only fences: the opaque bases below alias (see CUFHE_AMD_XPOSE)
#pragma unroll
    for (int r = 0; r < kRegs; r++) {
        *(uint32_t*)(wpos + 256 * r) = acc[r];
        *(uint32_t*)(wneg + 256 * r) = 0u - acc[r];
    }
    asm volatile("" ::: "memory");
    const char* rbase = tile + opaque(4 * (lane - alo + kN));
    uint32_t rot[kRegs];
#pragma unroll
    for (int r = 0; r < kRegs; r++) rot[r] = *(const uint32_t*)(rbase + 256 * r);   // all reads in flight together
#pragma unroll
    for (int r = 0; r < kRegs; r++) temp[r] = (rot[r] - acc[r] + decomp_offset()) ^ decomp_signmask();
}

// x (spectrum of one digit polynomial, layout C) times the two polynomials of one TRGSW row,
// read from the row buffer the workgroup staged in LDS: [out<2][q<8][lane<64][2] doubles
// One product into its accumulator: wide or narrow by the compile-time bound of the spectrum register (ntt_r4.h); LAST:
// the product that completes the sum reduces it on the way (fpf::mulmod_add: one operation more than a plain product,
// three fewer than a separate reduction in front of the inverse transform)
template <class SPEC, int R, bool LAST>
__device__ __forceinline__ void product_into(double& acc, double x, double w)
{
    constexpr bool wide = r4::needs_wide(SPEC::in().v[R]);
    if constexpr (LAST) acc = wide ? fpf::mulmod_add_wide(x, w, acc) : fpf::mulmod_add(x, w, acc);
    else acc += wide ? fpf::mulmod_wide(x, w) : fpf::mulmod(x, w);
}
template <class SPEC, bool LAST, int Q>
__device__ __forceinline__ void pointwise_piece(double (&A0)[kRegs], double (&A1)[kRegs], const double (&x)[kRegs], const double2& b)
{
    if constexpr (Q < 8) {
        product_into<SPEC, 2 * Q, LAST>(A0[2 * Q], x[2 * Q], b.x);
        product_into<SPEC, 2 * Q + 1, LAST>(A0[2 * Q + 1], x[2 * Q + 1], b.y);
    } else {
        product_into<SPEC, 2 * (Q - 8), LAST>(A1[2 * (Q - 8)], x[2 * (Q - 8)], b.x);
        product_into<SPEC, 2 * (Q - 8) + 1, LAST>(A1[2 * (Q - 8) + 1], x[2 * (Q - 8) + 1], b.y);
    }
}
template <class SPEC, bool LAST, int Q = 0>
__device__ __forceinline__ void pointwise_pieces(double (&A0)[kRegs], double (&A1)[kRegs], const double (&x)[kRegs],
                                                 double2 (&b)[16], const char* row_lane)
{
    // Software pipeline, pinned: the ds_read_b128 of piece q+D is issued before the products of
    // piece q (hipcc otherwise sinks every read to its use and waits for it there).  The
    // sched_barrier lets VALU/SALU instructions float but keeps DS reads on their side.
#ifndef CUFHE_AMD_BK_DEPTH
#define CUFHE_AMD_BK_DEPTH 3
#endif
    constexpr int D = CUFHE_AMD_BK_DEPTH;
    if constexpr (Q < 16) {
        if constexpr (Q + D < 16) b[Q + D] = *(const double2*)(row_lane + (Q + D) * 1024);
        __builtin_amdgcn_sched_barrier(0x0006);
        pointwise_piece<SPEC, LAST, Q>(A0, A1, x, b[Q]);
        __builtin_amdgcn_sched_barrier(0x0006);
        pointwise_pieces<SPEC, LAST, Q + 1>(A0, A1, x, b, row_lane);
    }
}
template <class SPEC, bool LAST>
__device__ __forceinline__ void pointwise_accumulate(double (&A0)[kRegs], double (&A1)[kRegs],
                                                     const double (&x)[kRegs], const char* row_lane)
{
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_NO_BK)
#pragma unroll
    for (int q = 0; q < 8; q++) {
        A0[2 * q] += fpf::mulmod_wide(x[2 * q], 1234567.0 + q);
        A0[2 * q + 1] += fpf::mulmod_wide(x[2 * q + 1], 7654321.0 + q);
        A1[2 * q] += fpf::mulmod_wide(x[2 * q], 2234567.0 + q);
        A1[2 * q + 1] += fpf::mulmod_wide(x[2 * q + 1], 8654321.0 + q);
    }
    (void)row_lane;
    return;
#endif
    constexpr int D = CUFHE_AMD_BK_DEPTH;
    double2 b[16];
#pragma unroll
    for (int q = 0; q < D; q++) b[q] = *(const double2*)(row_lane + q * 1024);
    pointwise_pieces<SPEC, LAST>(A0, A1, x, b, row_lane);
}

// The workgroup's row pipeline.  Row R (0 .. 6*steps-1) of the bootstrapping key is the
// 16 KiB block bk_ntt[R * 2048 ..]; it is copied by LDS-DMA into buffer R % 3, two 1 KiB
// pieces per wave, one row ahead of its use.  Every wave passes exactly one barrier per row:
//     s_waitcnt vmcnt(0)     this wave's pieces of row R have landed (issued a row ago)
//     s_barrier              => every wave's pieces of row R have landed
//     issue row R+1 into buffer (R+1) % 3
// Waves 0-3 ("early") take that barrier between the forward NTT of row R and its pointwise
// product; waves 4-7 ("late") take the SAME barrier two phases earlier in their own
// program, between stages 0-3 and the first transpose of the NTT of row R.  The late half
// therefore trails the early half by half a row for the whole kernel: on every SIMD (wave
// w and w+4 share one) a wave that waits on an LDS transpose sits beside a wave in a pure
// FP64 phase, instead of two waves in lock-step waiting together (measured: transposes
// cost 21 % of the kernel in lock-step).  Three buffers make this safe: buffer (R+1) % 3
// last held row R-2, whose last reader (a late wave's pointwise) precedes that wave's
// barrier R-1.
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_PHASES)
// timing-only: cycles a wave spends inside the row barrier (tools/br_phases.py)
#define CUFHE_AMD_DIAG_ARG , unsigned long long& diag_wait
#define CUFHE_AMD_DIAG_PASS , diag_wait
#define CUFHE_AMD_ROW_SYNC(R) { const unsigned long long t0_ = __builtin_readcyclecounter(); pipe.sync(R); diag_wait += __builtin_readcyclecounter() - t0_; }
#else
#define CUFHE_AMD_DIAG_ARG
#define CUFHE_AMD_DIAG_PASS
#define CUFHE_AMD_ROW_SYNC(R) pipe.sync(R);
#endif
struct RowPipe {
    const char* bk;          // NTT-domain key, bytes
    char* buf;               // LDS: kBkRowBuffers x kBkRowBytes
    int wave, lane, total_rows;
    bool late;
    __device__ __forceinline__ void issue(int R) const
    {
        if (R >= total_rows) return;
        const char* src = bk + (size_t)R * kBkRowBytes + lane * 16;
        char* dst = buf + (R % kBkRowBuffers) * kBkRowBytes;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int piece = 2 * wave + c;
            lds_dma16(src + piece * 1024, dst + piece * 1024);
        }
    }
    __device__ __forceinline__ void sync(int R) const
    {
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_NO_SYNC)
        (void)R;             // timing only: no row barrier, no key traffic -- what the waves do when nothing couples them
        return;
#endif
        lds_dma_wait_all();  // this wave's pieces of row R (issued a row ago) have landed
        __syncthreads();     // s_waitcnt lgkmcnt(0); s_barrier
        issue(R + 1);
    }
    __device__ __forceinline__ const char* row(int R) const
    {
        return buf + opaque((R % kBkRowBuffers) * kBkRowBytes + lane * 16);
    }
};

// The schedule of the step (ntt_r4.h): spectrum of a digit polynomial, the (k+1) l products per accumulator the last of
// which reduces the sum, the inverse transform of that
using BrSpectrum = r4::FwdDigits<(1 << (kBgbit - 1))>::Spectrum;
using BrSums = r4::PointwiseSum<BrSpectrum, kBkRows, true>;
using BrInverse = r4::Inverse<BrSums>;
static_assert(r4::valid(BrSpectrum::in()) && r4::valid(BrSums::in()) && r4::valid(BrInverse::Out::in()),
              "blind_rotate_kernel: the lazy-reduction schedule of the CMux step exceeds the FP64 mantissa");

// one component j: rotate/subtract/decompose, then l forward NTTs, each multiplied into
// both accumulators (include/gatebootstrapping_gpu.cuh:153-224).  LAST_COMPONENT: its last row completes both sums.
template <bool LAST_COMPONENT>
__device__ __forceinline__ void cmux_component(double (&A0)[kRegs], double (&A1)[kRegs],
                                               const uint32_t (&accj)[kRegs], const WaveCtx& ctx,
                                               char* tile, int lane, uint32_t abar,
                                               const RowPipe& pipe, int first_row, const TuFwdPinned& tuf CUFHE_AMD_DIAG_ARG)
{
    constexpr int kDigitMax = 1 << (kBgbit - 1);
    uint32_t temp[kRegs];
    rotate_sub(temp, accj, tile, lane, abar);
#pragma unroll 1
    for (int d = 0; d < kL; d++) {
        const uint32_t pos = 32 - (d + 1) * kBgbit;  // v_bfe_i32: the sign-extended Bgbit-wide field at pos
        double x[kRegs];
#pragma unroll
        for (int r = 0; r < kRegs; r++)
            x[r] = (double)(int32_t)__builtin_amdgcn_sbfe(temp[r], pos, (uint32_t)kBgbit);
        double twb[kTbCount];
        ntt_forward_digits_a_r4<kDigitMax, true>(x, ctx, &tuf, &twb);
        if (pipe.late) CUFHE_AMD_ROW_SYNC(first_row + d)
        ntt_forward_digits_bc_r4<kDigitMax, false, true>(x, ctx, twb);
        if (!pipe.late) CUFHE_AMD_ROW_SYNC(first_row + d)
        if (LAST_COMPONENT && d == kL - 1) pointwise_accumulate<BrSpectrum, true>(A0, A1, x, pipe.row(first_row + d));
        else pointwise_accumulate<BrSpectrum, false>(A0, A1, x, pipe.row(first_row + d));
    }
}

template <bool TWB_LOADED>
__device__ __forceinline__ void inverse_and_add(double (&A)[kRegs], uint32_t (&accj)[kRegs], const WaveCtx& ctx,
                                                const double (&twc)[kTcCount], double (&twb)[kTbCount])
{
    ntt_inverse_r4_tw<BrSums, false, TWB_LOADED>(A, ctx, twc, twb);
    lift_add<BrInverse::Out>(accj, A);           // centred lift, :258-281
}

// descs[count]: in0/in1 are lvl0 TLWEs, out is a lvl1 TLWE (N+1 words, sample extract at
// index 0).  steps < n is only used by the parity tests; acc_dump (optional) receives the
// raw accumulator (2N words per rotation).  All 8 waves of a workgroup walk the key in
// lock-step (one barrier per TRGSW row); waves past `count` only serve the row pipeline.
// `active` (8 or 4) is the number of waves per workgroup that own a rotation: with 4, every SIMD
// runs ONE rotation instead of two and a round of the grid takes about 11 ms instead of 19 -- the
// shape for the tail of a launch that does not fill a second round (capi.hip: launch_blind_rotate).
__global__ __launch_bounds__(kBrThreads, 2) void blind_rotate_kernel(
    const LinDesc* __restrict__ descs, int count, const double* __restrict__ bk_ntt,
    const NttTables* __restrict__ gt, int steps, uint32_t* __restrict__ acc_dump, int active)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tabs = (double*)(smem + kBrLdsTables);
    load_packed_tables_to_lds(tabs, gt);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * active + wave;
    char* tile = smem + kBrLdsTiles + wave * kTileBytes;
    uint16_t* abar_lds = (uint16_t*)(smem + kBrLdsAbar + wave * kAbarBytes);
    const WaveCtx ctx = make_wave_ctx_packed(smem, kBrLdsTiles + wave * kTileBytes, kBrLdsTables, gt, lane);
    const RowPipe pipe{(const char*)bk_ntt, smem + kBrLdsBk, wave, lane, steps * kBkRows, wave >= kBrWavesPerBlock / 2};
    pipe.issue(0);
    if (wave >= active || g >= count) {
        // no rotation for this wave (tail of the batch): it only keeps its share of the row
        // pipeline going -- one barrier and two LDS-DMA pieces per row -- and computes nothing
        __syncthreads();
#pragma unroll 1
        for (int R = 0; R < pipe.total_rows; R++) pipe.sync(R);
        return;
    }

    const LinDesc d = descs[g];
    // pre-add (gate linear part) and modulus switch, :316-345
    uint32_t bword = 0;
#pragma unroll
    for (int rr = 0; rr < 10; rr++) {
        const int i = lane + 64 * rr;
        uint32_t c = 0;
        if (i <= kLvl0N) c = (uint32_t)d.ca * d.in0[i] + (uint32_t)d.cb * d.in1[i];
        if (rr == 9) bword = c;
        if (i < kLvl0N) abar_lds[i] = (uint16_t)((c + (1u << (32 - 2 - kNbit))) >> (32 - 1 - kNbit));
    }
    bword = __builtin_amdgcn_readlane(bword, kLvl0N - 64 * 9) + d.off;
    const uint32_t bbar = 2 * kN - (bword >> (32 - 1 - kNbit));

    // RotatedTestVector, :29-52
    uint32_t acc0[kRegs], acc1[kRegs];
#pragma unroll
    for (int r = 0; r < kRegs; r++) {
        const uint32_t e = lane + 64 * r;
        acc0[r] = 0;
        const bool neg = (bbar != 2 * kN) && ((e < (bbar & (kN - 1))) != ((bbar >> kNbit) != 0));
        acc1[r] = neg ? 0u - kMu : kMu;
    }
    __syncthreads();          // tables staged; abar list visible (own wave only, but cheap)
    TuFwdPinned tuf;
    tuf.load(gt);
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_PHASES)
    unsigned long long diag_wait = 0;
    const unsigned long long diag_t0 = __builtin_readcyclecounter();
#endif

#pragma unroll 1
    for (int i = 0; i < steps; i++) {
        // abar = 0 needs no special case: all digits are zero and the step adds nothing
        const uint32_t abar = __builtin_amdgcn_readfirstlane((uint32_t)abar_lds[i]);
        double A0[kRegs], A1[kRegs], inv_twc[kTcCount], inv_twb[kTbCount];
#pragma unroll
        for (int r = 0; r < kRegs; r++) { A0[r] = 0.0; A1[r] = 0.0; }
        // six products per accumulator, the last one reducing the sum (BrSums)
        cmux_component<false>(A0, A1, acc0, ctx, tile, lane, abar, pipe, i * kBkRows, tuf CUFHE_AMD_DIAG_PASS);
        cmux_component<true>(A0, A1, acc1, ctx, tile, lane, abar, pipe, i * kBkRows + kL, tuf CUFHE_AMD_DIAG_PASS);
        // the two inverse transforms share their per-lane twiddles: fetched once
        load_packed(inv_twc, ctx.tc_inv);
        inverse_and_add<false>(A0, acc0, ctx, inv_twc, inv_twb);     // fetches the stage 7-4 twiddles ...
        inverse_and_add<true>(A1, acc1, ctx, inv_twc, inv_twb);      // ... which the second transform re-uses
    }

    if (acc_dump) {
        uint32_t* o = acc_dump + (size_t)g * 2 * kN;
#pragma unroll
        for (int r = 0; r < kRegs; r++) {
            o[lane + 64 * r] = acc0[r];
            o[kN + lane + 64 * r] = acc1[r];
        }
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_PHASES)
        if (lane == 0) {      // overwrites the first words of this rotation's dump: timing only
            ((unsigned long long*)o)[0] = __builtin_readcyclecounter() - diag_t0;
            ((unsigned long long*)o)[1] = diag_wait;
        }
#endif
    }
    if (d.out) {
        // __SampleExtractIndex__<P,0>: out[0] = a[0], out[m] = -a[N-m], out[N] = b[0]
        uint32_t* o = d.out;
#pragma unroll
        for (int r = 0; r < kRegs; r++) {
            const int e = lane + 64 * r;
            if (e == 0) { o[0] = acc0[r]; o[kN] = acc1[r]; }
            else o[kN - e] = 0u - acc0[r];
        }
    }
}

// ----------------------------------------------------------------------------------
// Key switch lvl1 -> lvl0 with the linear pre-add fused (IdentityKeySwitchPreAdd,
// include/keyswitch_gpu.cuh:136-188; KeySwitchFromTLWE :83-134 is the ca=1, cb=0 case).
//
// The reference gives every ciphertext its own pass over the key (thread i walks
// ksk[j][k][|val|-1][i] for all j, k): ~15.5 MB of table per ciphertext.  A first version of
// this kernel (one wave per ciphertext reading the rows straight from L2) ran at 27 TB/s of
// L2 traffic -- the L2 bandwidth limit -- and still took 3.1 ms for 4096 ciphertexts.
// So the table is shared instead: a workgroup of 16 waves handles 16 ciphertexts and
// walks j in lock-step; the 16 candidate rows of one j (t = 8 levels x 2 values, contiguous
// 40 KiB in the padded device layout [j][k][v][640]) are copied ONCE into LDS by LDS-DMA,
// two steps ahead (3 buffers), and every wave adds or subtracts the 8 rows its own digits
// select (wave-uniform choice per digit).  L2 traffic drops 16x.
// Lane L owns 16-byte pieces L and L+64 and the 8-byte piece L behind them (words 4L..4L+3,
// 256+4L.. and 512+2L, 513+2L): ten words, two ds_read_b128 and one ds_read_b64 per row.
// What a launch of 4096 costs (1.04 ms; round 5: 1.33): the 16 waves read 6 of their 8 rows
// per step, 245 KB out of LDS per step and CU -- 0.8 ms at 128 bytes per cycle -- next to the
// 40 KiB the DMA writes into it; the table pipeline alone (no digits) takes 0.40 ms, the
// digits without it 1.0 ms (diagnostic builds KS_NO_DIGITS / KS_NO_DMA, tools/ks_floor.py,
// profiles/r06_keyswitch.md).
// ----------------------------------------------------------------------------------
constexpr int kKsWaves = 16;
constexpr int kKsThreads = 64 * kKsWaves;                        // 1024
constexpr int kKsRowPad = 640;                                   // words per padded row
constexpr int kKsPieces = 3;                                     // 16-byte pieces per lane (the workgroup-per-ciphertext kernels)
constexpr int kKsStepBytes = kKsT * kKsNumBase * kKsRowPad * 4;  // 40960: all rows of one j
constexpr int kKsBuffers = 3;
constexpr int kKsDigitSteps = 1024;                              // digit words a wave keeps in LDS: the steps of one workgroup
constexpr int kKsLdsDigits = kKsWaves * kKsDigitSteps * 2;       // u16 digit words: 32768

// The kernel is written once over the SHAPE of a key switch (basebit = 2 everywhere):
//   Desc                      what a launch is made of (LinDesc / LinDesc64)
//   kn, t                     values a'_j per ciphertext, digits per value
//   row_pad, n_out            words per padded row of the table (512 or 640), index of the last output word (b')
//   digit_word(d, j)          the t digit fields of a'_j, most significant first, in the top bits of a 16-bit word
//   bprime(d)                 the start of output word n_out
// KsShapeDefault: lvl10 of the BASELINE set (here); KsShapeCggi16 (kernels_ps.hip.h); KsShapeLvl2: lvl20 (kernels_ks2.hip.h).
struct KsShapeDefault {
    using Desc = LinDesc;
    static constexpr int kn = kN, t = kKsT, row_pad = kKsRowPad, n_out = kLvl0N;
    // iksoffsetgen + roundoffset, include/keyswitch_gpu.cuh:13-23,92-98; only the top t*basebit = 16 bits of a'_j + offset carry digits
    static constexpr uint32_t koff()
    {
        uint32_t o = 1u << (32 - (1 + kKsBasebit * kKsT));
        for (int i = 1; i <= kKsT; i++) o += ((1u << kKsBasebit) / 2) << (32 - i * kKsBasebit);
        return o;
    }
    static __device__ __forceinline__ uint32_t digit_word(const Desc& d, int j)
    {
        return ((uint32_t)d.ca * d.in0[j] + (uint32_t)d.cb * d.in1[j] + koff()) >> 16;
    }
    static __device__ __forceinline__ uint32_t bprime(const Desc& d) { return (uint32_t)d.ca * d.in0[kn] + (uint32_t)d.cb * d.in1[kn] + d.off; }
};
template <class S>
struct KsDims {
    static constexpr int pairs = (S::row_pad - 512) / 128;              // 8-byte pieces per lane behind the two 16-byte ones: 0 or 1
    static constexpr int words = 8 + 2 * pairs;                         // output words per lane
    static constexpr int step_bytes = S::t * kKsNumBase * S::row_pad * 4;       // all rows of one j
    static constexpr int dma_pieces = step_bytes / 1024;                // LDS-DMA pieces of 1 KiB per step
    static constexpr int dma_more = dma_pieces - 2 * kKsWaves;          // waves that move three pieces (the others two)
    static constexpr int lds_bytes = kKsLdsDigits + kKsBuffers * step_bytes;
    static constexpr int min_slices = (S::kn + kKsDigitSteps - 1) / kKsDigitSteps;
    static_assert(S::row_pad == 512 || S::row_pad == 640, "a lane owns two quads and at most one pair of a row");
    static_assert(step_bytes % 1024 == 0 && dma_more >= 0 && dma_more <= kKsWaves, "two or three DMA pieces per wave and step");
    static_assert(S::t * kKsBasebit <= 16 && S::n_out < S::row_pad, "digit word of 16 bits; the output fits a row");
    static_assert(lds_bytes <= 160 * 1024, "digit words + three step buffers fit the CU's LDS");
};
static_assert(KsDims<KsShapeDefault>::step_bytes == kKsStepBytes && KsDims<KsShapeDefault>::lds_bytes == 155648, "BASELINE shape");

// One digit of one key-switch step on one wave: f = val + 2 (wave-uniform, in an SGPR) selects +row(v=2) [f 0], +row(v=1) [f 1],
// nothing [f 2] or -row(v=1) [f 3]; the row lies at a compile-time offset from the lane's LDS addresses and is added to the
// lane's eight or ten sums IN PLACE.  The three-way choice is spelled as scalar branches INSIDE asm blocks, so that the compiler sees
// straight-line code: written as `if (f == 3) res -= r; else res += r;` the structurised control flow gave every arm fresh result
// registers and copied them back (6 v_mov_b64 per digit) and the row offset went through a VGPR (3 v_add per digit) -- 21 vector
// instructions per digit where 10 do the work.  And the rows of ALL digits of a step are requested before the first is added (one
// wait): waited for one by one they are a chain of eight dependent LDS round trips per step, which with the scalar branching around
// them was what a step took (0.75 us even for a single ciphertext per workgroup; the table pipeline alone: 0.4 us).
// profiles/r06_keyswitch.md
typedef uint32_t ks_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t ks_u32x2 __attribute__((ext_vector_type(2)));
struct KsRow { ks_u32x4 a, b; ks_u32x2 c; };
// ... the loads: issued, NOT waited for
template <int ROW1, int ROW2, int PAIRS>
__device__ __forceinline__ void ks_digit_load(const uint32_t f, const uint32_t a0, const uint32_t a1, const uint32_t a2, KsRow& r)
{
    if constexpr (PAIRS) {
        asm volatile(
            "s_cmp_eq_u32 %3, 2\n\t"
            "s_cbranch_scc1 .Lks_ld_end_%=\n\t"
            "s_cmp_eq_u32 %3, 0\n\t"
            "s_cbranch_scc1 .Lks_ld_two_%=\n\t"
            "ds_read_b128 %0, %4 offset:%7\n\t"
            "ds_read_b128 %1, %5 offset:%7\n\t"
            "ds_read_b64 %2, %6 offset:%7\n\t"
            "s_branch .Lks_ld_end_%=\n"
            ".Lks_ld_two_%=:\n\t"
            "ds_read_b128 %0, %4 offset:%8\n\t"
            "ds_read_b128 %1, %5 offset:%8\n\t"
            "ds_read_b64 %2, %6 offset:%8\n"
            ".Lks_ld_end_%=:"
            : "=&v"(r.a), "=&v"(r.b), "=&v"(r.c)
            : "s"(f), "v"(a0), "v"(a1), "v"(a2), "n"(ROW1), "n"(ROW2)
            : "memory", "scc");
    } else {
        asm volatile(
            "s_cmp_eq_u32 %2, 2\n\t"
            "s_cbranch_scc1 .Lks_ld_end_%=\n\t"
            "s_cmp_eq_u32 %2, 0\n\t"
            "s_cbranch_scc1 .Lks_ld_two_%=\n\t"
            "ds_read_b128 %0, %3 offset:%5\n\t"
            "ds_read_b128 %1, %4 offset:%5\n\t"
            "s_branch .Lks_ld_end_%=\n"
            ".Lks_ld_two_%=:\n\t"
            "ds_read_b128 %0, %3 offset:%6\n\t"
            "ds_read_b128 %1, %4 offset:%6\n"
            ".Lks_ld_end_%=:"
            : "=&v"(r.a), "=&v"(r.b)
            : "s"(f), "v"(a0), "v"(a1), "n"(ROW1), "n"(ROW2)
            : "memory", "scc");
    }
}
// ... the additions or subtractions.  (Without a branch for the sign -- res += row ^ s through v_xad_u32, s = 0 or ~0, the
// subtractions counted in the prologue and added to every word at the end -- a launch of up to 1024 ciphertexts is 7 % faster and
// one of 4096 is 4 % slower: profiles/r06_keyswitch_xad_experiment.patch.)
template <int PAIRS>
__device__ __forceinline__ void ks_digit_acc(const uint32_t f, const KsRow& r, uint32_t (&res)[8 + 2 * PAIRS])
{
    if constexpr (PAIRS) {
#define CUFHE_AMD_KS_ALL(OP)                                                                                               \
    OP " %0, %0, %11\n\t" OP " %1, %1, %12\n\t" OP " %2, %2, %13\n\t" OP " %3, %3, %14\n\t" OP " %4, %4, %15\n\t"             \
    OP " %5, %5, %16\n\t" OP " %6, %6, %17\n\t" OP " %7, %7, %18\n\t" OP " %8, %8, %19\n\t" OP " %9, %9, %20\n"
        asm volatile(
            "s_cmp_eq_u32 %10, 2\n\t"
            "s_cbranch_scc1 .Lks_acc_end_%=\n\t"
            "s_cmp_eq_u32 %10, 3\n\t"
            "s_cbranch_scc1 .Lks_acc_sub_%=\n\t"
            CUFHE_AMD_KS_ALL("v_add_u32_e32")
            "\ts_branch .Lks_acc_end_%=\n"
            ".Lks_acc_sub_%=:\n\t"
            CUFHE_AMD_KS_ALL("v_sub_u32_e32")
            ".Lks_acc_end_%=:"
            : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]), "+v"(res[5]), "+v"(res[6]), "+v"(res[7]), "+v"(res[8]), "+v"(res[9])
            : "s"(f), "v"(r.a.x), "v"(r.a.y), "v"(r.a.z), "v"(r.a.w), "v"(r.b.x), "v"(r.b.y), "v"(r.b.z), "v"(r.b.w), "v"(r.c.x), "v"(r.c.y)
            : "scc");
#undef CUFHE_AMD_KS_ALL
    } else {
#define CUFHE_AMD_KS_ALL(OP)                                                                                               \
    OP " %0, %0, %9\n\t" OP " %1, %1, %10\n\t" OP " %2, %2, %11\n\t" OP " %3, %3, %12\n\t" OP " %4, %4, %13\n\t"              \
    OP " %5, %5, %14\n\t" OP " %6, %6, %15\n\t" OP " %7, %7, %16\n"
        asm volatile(
            "s_cmp_eq_u32 %8, 2\n\t"
            "s_cbranch_scc1 .Lks_acc_end_%=\n\t"
            "s_cmp_eq_u32 %8, 3\n\t"
            "s_cbranch_scc1 .Lks_acc_sub_%=\n\t"
            CUFHE_AMD_KS_ALL("v_add_u32_e32")
            "\ts_branch .Lks_acc_end_%=\n"
            ".Lks_acc_sub_%=:\n\t"
            CUFHE_AMD_KS_ALL("v_sub_u32_e32")
            ".Lks_acc_end_%=:"
            : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]), "+v"(res[5]), "+v"(res[6]), "+v"(res[7])
            : "s"(f), "v"(r.a.x), "v"(r.a.y), "v"(r.a.z), "v"(r.a.w), "v"(r.b.x), "v"(r.b.y), "v"(r.b.z), "v"(r.b.w)
            : "scc");
#undef CUFHE_AMD_KS_ALL
    }
}
template <int I, int N, class F>
__device__ __forceinline__ void ks_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ks_for<I + 1, N>(f);
    }
}
// the t digits of a'_j (dj: its digit bits, most significant digit first): all rows requested (not waited for) / all additions
__device__ __forceinline__ uint32_t ks_field(const uint32_t dj, const int k) { return (dj >> (16 - (k + 1) * kKsBasebit)) & ((1u << kKsBasebit) - 1); }
template <class S>
__device__ __forceinline__ void ks_load_all(const uint32_t dj, const uint32_t (&pb)[3], KsRow (&r)[S::t])
{
    ks_for<0, S::t>([&](auto kc) {
        constexpr int K = decltype(kc)::value;
        ks_digit_load<(K * kKsNumBase) * (S::row_pad * 4), (K * kKsNumBase + 1) * (S::row_pad * 4), KsDims<S>::pairs>(ks_field(dj, K), pb[0], pb[1], pb[2], r[K]);
    });
}
template <class S>
__device__ __forceinline__ void ks_acc_all(const uint32_t dj, const KsRow (&r)[S::t], uint32_t (&res)[KsDims<S>::words])
{
    ks_for<0, S::t>([&](auto kc) { ks_digit_acc<KsDims<S>::pairs>(ks_field(dj, decltype(kc)::value), r[decltype(kc)::value], res); });
}

// the outputs of a launch that cuts j into runs: zeroed first (the runs add their partial sums with atomics)
template <class S>
__global__ __launch_bounds__(256) void keyswitch_zero_kernel(const typename S::Desc* __restrict__ descs, int count)
{
    const int g = blockIdx.x;
    if (g >= count) return;
    uint32_t* out = descs[g].out;
    for (int i = threadIdx.x; i <= S::n_out; i += blockDim.x) out[i] = 0u;
}

// per_wg (1..16): ciphertexts per workgroup.  Waves at and above per_wg only move table pieces and keep the barriers: a
// launch of fewer than 4096 ciphertexts then still covers every CU, and a step carries fewer row reads and additions.
// slices (a power of two, at most 64, at least kn / 1024): the steps of j are cut into `slices` runs and workgroup i takes run
// i % slices for the ciphertexts of group i / slices (workgroups of one XCD -- i % 8 -- then walk the same part of the table).  With
// slices > 1 the partial sums are added into d.out with atomics, which keyswitch_zero_kernel has zeroed before: a launch of 2048
// ciphertexts is then 128 groups x 2 runs of 512 steps with all 16 waves live instead of 256 workgroups x 1024 steps with 8, one of
// 256 is 16 groups x 16 runs of 64 steps (profiles/r06_keyswitch.md).
template <class S>
__global__ __launch_bounds__(kKsThreads) void keyswitch_kernel(
    const typename S::Desc* __restrict__ descs, int count, const uint32_t* __restrict__ ksk_padded, int per_wg, int slices)
{
    using K = KsDims<S>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int slice = blockIdx.x % slices;
    const int steps = S::kn / slices, j_lo = slice * steps;    // this workgroup's run of j: [j_lo, j_lo + steps); t below = j - j_lo
    int g = (blockIdx.x / slices) * per_wg + wave;
    const bool live = wave < per_wg && g < count;
    if (!live) g = count - 1;
    const typename S::Desc d = descs[g];
    uint16_t* dig = (uint16_t*)smem + wave * kKsDigitSteps;
    char* bufs = smem + kKsLdsDigits;

    // LDS-DMA of step t in pieces of 1 KiB: the first dma_more waves move three, the others two
    const bool three = wave < K::dma_more;
    const int piece0 = 2 * wave + (three ? wave : K::dma_more);
    auto issue = [&](int t) {
        if (t >= steps) return;
        const char* src = (const char*)ksk_padded + (size_t)(j_lo + t) * K::step_bytes + lane * 16;
        char* dst = bufs + (t % kKsBuffers) * K::step_bytes;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (c == 2 && !three) break;
#if !(defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_KS_NO_DMA))
            lds_dma16(src + (piece0 + c) * 1024, dst + (piece0 + c) * 1024);
#endif
        }
    };
    issue(0);
    issue(1);

    for (int t = lane; t < steps; t += 64) dig[t] = (uint16_t)S::digit_word(d, j_lo + t);
    const uint32_t bprime = slice == 0 ? S::bprime(d) : 0u;     // the start of word n_out, in the first run only

    // Lane L owns 16-byte pieces L and L + 64 (words 4 L .. 4 L + 3 and 256 + 4 L ..) and, in a row of 640 words, the 8-byte piece L
    // behind them (words 512 + 2 L, 513 + 2 L): eight or ten words, no lane idles on a third quad.
    uint32_t res[K::words];
#pragma unroll
    for (int m = 0; m < K::words; m++) res[m] = 0;
    {
        constexpr int n = S::n_out, m = n < 256 ? n % 4 : n < 512 ? 4 + (n - 256) % 4 : 8 + (n - 512) % 2;
        constexpr int owner = n < 256 ? n / 4 : n < 512 ? (n - 256) / 4 : (n - 512) / 2;
        if (lane == owner) res[m] = bprime;
    }
    // per-buffer LDS addresses of the pieces kept in VGPRs: a row is then "VGPR + immediate"
    uint32_t pbase[kKsBuffers][3];
#pragma unroll
    for (int bi = 0; bi < kKsBuffers; bi++) {
        const uint32_t buf = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(bufs + bi * K::step_bytes);
        pbase[bi][0] = buf + lane * 16;
        pbase[bi][1] = buf + 1024 + lane * 16;
        pbase[bi][2] = buf + 2048 + lane * 8;
    }

    __syncthreads();          // digit words visible; the prologue's plain loads have drained vmcnt
    // One step t: counted wait + barrier, issue step t+2, apply the digits of a'_j.  The waves of a SIMD (wave w runs on SIMD w % 4)
    // come out of the barrier together and would all request rows (scalar and LDS work), then all wait, then all add (vector
    // work), one pipe busy at a time.  So every second wave of a SIMD runs one phase behind: it requests the rows of step t at the
    // END of the step's interval and adds them at the start of the next one, while its neighbours request theirs.
    auto wait_and_issue = [&](int t) {
        // The pieces of step t+1 (this wave's newest 3 or 2 DMAs) stay in flight across the
        // barrier, only step t must have landed.  lgkmcnt(0): this wave has finished reading
        // step t-1, whose buffer step t+2 is about to overwrite.
        if (t + 1 >= steps) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (three) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        issue(t + 2);
    };
    static_assert(kKsBuffers == 3, "the loops over t are unrolled by the number of buffers");
    const int whole = steps - steps % 3;       // steps is a power of two: one or two steps follow the unrolled loop
    KsRow r[S::t];
    // (with at most two live waves per SIMD -- per_wg <= 8 -- the shift costs more than it hides: 0.90 against 0.86 ms per 2048)
    if (per_wg <= 8 || !((wave >> 2) & 1)) {
        auto step = [&](int t, const uint32_t (&pb)[3]) {
            wait_and_issue(t);
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_KS_NO_DIGITS)
            return;                                               // timing only: the table pipeline without the digits
#endif
            if (!live) return;                                    // wave-uniform: this wave only serves the table pipeline
            const uint32_t dj = __builtin_amdgcn_readfirstlane((uint32_t)dig[t]);
            ks_load_all<S>(dj, pb, r);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ks_acc_all<S>(dj, r, res);
        };
#pragma unroll 1
        for (int t = 0; t < whole; t += 3) {
            step(t, pbase[0]);
            step(t + 1, pbase[1]);
            step(t + 2, pbase[2]);
        }
        step(whole, pbase[0]);
        if (whole + 1 < steps) step(whole + 1, pbase[1]);
    } else {
        auto step = [&](int t, const uint32_t (&pb)[3]) {
            wait_and_issue(t);                                    // its lgkmcnt(0): the rows of step t-1 are in r
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_KS_NO_DIGITS)
            return;                                               // timing only: the table pipeline without the digits
#endif
            if (!live) return;
            if (t > 0) ks_acc_all<S>(__builtin_amdgcn_readfirstlane((uint32_t)dig[t - 1]), r, res);
            ks_load_all<S>(__builtin_amdgcn_readfirstlane((uint32_t)dig[t]), pb, r);
        };
#pragma unroll 1
        for (int t = 0; t < whole; t += 3) {
            step(t, pbase[0]);
            step(t + 1, pbase[1]);
            step(t + 2, pbase[2]);
        }
        step(whole, pbase[0]);
        if (whole + 1 < steps) step(whole + 1, pbase[1]);
        if (live) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ks_acc_all<S>(__builtin_amdgcn_readfirstlane((uint32_t)dig[steps - 1]), r, res);
        }
    }
    if (!live) return;
    uint32_t* o = d.out;                                     // 4-byte aligned only (ciphertexts packed at n_out + 1 words)
    auto put = [&](int i, uint32_t v) {
        if (i > S::n_out) return;
        if (slices == 1) o[i] = v;
        else atomicAdd(&o[i], v);
    };
#pragma unroll
    for (int m = 0; m < 4; m++) { put(4 * lane + m, res[m]); put(256 + 4 * lane + m, res[4 + m]); }
    if constexpr (K::pairs) { put(512 + 2 * lane, res[8]); put(513 + 2 * lane, res[9]); }
}

// Low-latency key switch: one workgroup (16 waves) per ciphertext, wave w takes the 64 values
// a'_j, j in [64 w, 64 w + 64), reads its rows straight from L2 (eight rows in flight) and the
// 16 partial sums are added through LDS.  Same words as keyswitch_kernel; used for small
// launches, where sharing the table between ciphertexts buys nothing.
__global__ __launch_bounds__(kKsThreads) void keyswitch_wg_kernel(
    const LinDesc* __restrict__ descs, int count, const uint32_t* __restrict__ ksk_padded)
{
    __shared__ uint32_t part[kKsWaves][kKsRowPad];
    __shared__ uint16_t dig[kN];
    __shared__ uint32_t bprime_s;
    const int g = blockIdx.x;
    if (g >= count) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const LinDesc d = descs[g];
    uint32_t koff = 1u << (32 - (1 + kKsBasebit * kKsT));
    for (int i = 1; i <= kKsT; i++) koff += ((1u << kKsBasebit) / 2) << (32 - i * kKsBasebit);
    for (int j = tid; j < kLvl1Words; j += kKsThreads) {
        const uint32_t v = (uint32_t)d.ca * d.in0[j] + (uint32_t)d.cb * d.in1[j];
        if (j == kN) bprime_s = v + d.off;
        else dig[j] = (uint16_t)((v + koff) >> 16);
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
    for (int jj = 0; jj < kN / kKsWaves; jj++) {
        const int j = wave * (kN / kKsWaves) + jj;
        const uint32_t dj = __builtin_amdgcn_readfirstlane((uint32_t)dig[j]);
        int val[kKsT];
        uint4 row[kKsT][kKsPieces];
#pragma unroll
        for (int k = 0; k < kKsT; k++) {
            val[k] = (int)((dj >> (16 - (k + 1) * kKsBasebit)) & ((1u << kKsBasebit) - 1)) - (1 << (kKsBasebit - 1));
            const int v = val[k] > 0 ? val[k] : -val[k];
            const uint4* r = base + ((size_t)(j * kKsT + k) * kKsNumBase + (v ? v - 1 : 0)) * kRowPieces;
#pragma unroll
            for (int m = 0; m < kKsPieces; m++) row[k][m] = r[piece[m]];
        }
#pragma unroll
        for (int k = 0; k < kKsT; k++) {
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

// Lowest-latency key switch for a handful of ciphertexts: kKsSplit workgroups per ciphertext, each
// taking 128 values of j (8 per wave) and adding its partial sum into the output with 32-bit
// atomics (sums mod 2^32 are order-free: same words).  The output is zeroed by
// keyswitch_split_zero_kernel first; the workgroup of j = 0 adds b'.
constexpr int kKsSplit = 8;
__global__ __launch_bounds__(256) void keyswitch_split_zero_kernel(const LinDesc* __restrict__ descs, int count)
{
    const int g = blockIdx.x;
    if (g >= count) return;
    uint32_t* out = descs[g].out;
    for (int i = threadIdx.x; i <= kLvl0N; i += blockDim.x) out[i] = 0u;
}
__global__ __launch_bounds__(kKsThreads) void keyswitch_split_kernel(
    const LinDesc* __restrict__ descs, int count, const uint32_t* __restrict__ ksk_padded)
{
    __shared__ uint32_t part[kKsWaves][kKsRowPad];
    __shared__ uint16_t dig[kN / kKsSplit];
    const int g = blockIdx.x / kKsSplit, sp = blockIdx.x % kKsSplit;
    if (g >= count) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const LinDesc d = descs[g];
    uint32_t koff = 1u << (32 - (1 + kKsBasebit * kKsT));
    for (int i = 1; i <= kKsT; i++) koff += ((1u << kKsBasebit) / 2) << (32 - i * kKsBasebit);
    constexpr int kJ = kN / kKsSplit;                          // 128 values of j per workgroup
    const int j0 = sp * kJ;
    if (tid < kJ) {
        const uint32_t v = (uint32_t)d.ca * d.in0[j0 + tid] + (uint32_t)d.cb * d.in1[j0 + tid];
        dig[tid] = (uint16_t)((v + koff) >> 16);
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
    for (int jj = 0; jj < kJ / kKsWaves; jj++) {
        const int jl = wave * (kJ / kKsWaves) + jj;
        const int j = j0 + jl;
        const uint32_t dj = __builtin_amdgcn_readfirstlane((uint32_t)dig[jl]);
        int val[kKsT];
        uint4 row[kKsT][kKsPieces];
#pragma unroll
        for (int k = 0; k < kKsT; k++) {
            val[k] = (int)((dj >> (16 - (k + 1) * kKsBasebit)) & ((1u << kKsBasebit) - 1)) - (1 << (kKsBasebit - 1));
            const int v = val[k] > 0 ? val[k] : -val[k];
            const uint4* r = base + ((size_t)(j * kKsT + k) * kKsNumBase + (v ? v - 1 : 0)) * kRowPieces;
#pragma unroll
            for (int m = 0; m < kKsPieces; m++) row[k][m] = r[piece[m]];
        }
#pragma unroll
        for (int k = 0; k < kKsT; k++) {
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
        uint32_t v = 0u;
        if (i == kLvl0N && sp == 0) v = (uint32_t)d.ca * d.in0[kN] + (uint32_t)d.cb * d.in1[kN] + d.off;
#pragma unroll
        for (int w = 0; w < kKsWaves; w++) v += part[w][i];
        atomicAdd(d.out + i, v);
    }
}

// __SampleExtractIndex__<P,0> on TRLWEs in global memory: trlwe[count][2N] -> tlwe1[count][N+1]
__global__ __launch_bounds__(256) void sample_extract_kernel(uint32_t* __restrict__ tlwe1,
                                                             const uint32_t* __restrict__ trlwe, int count)
{
    for (int g = blockIdx.x; g < count; g += gridDim.x) {
        const uint32_t* in = trlwe + (size_t)g * 2 * kN;
        uint32_t* o = tlwe1 + (size_t)g * kLvl1Words;
        for (int m = threadIdx.x; m <= kN; m += blockDim.x)
            o[m] = (m == kN) ? in[kN] : (m == 0 ? in[0] : 0u - in[kN - m]);
    }
}

// the same with one descriptor per TRLWE: in0 = trlwe (2N words), out = lvl1 TLWE (N + 1 words)
__global__ __launch_bounds__(256) void sample_extract_desc_kernel(const LinDesc* __restrict__ descs, int count)
{
    for (int g = blockIdx.x; g < count; g += gridDim.x) {
        const uint32_t* in = descs[g].in0;
        uint32_t* o = descs[g].out;
        for (int m = threadIdx.x; m <= kN; m += blockDim.x)
            o[m] = (m == kN) ? in[kN] : (m == 0 ? in[0] : 0u - in[kN - m]);
    }
}

// ----------------------------------------------------------------------------------
// CMUX against a caller-supplied TRGSW in the NTT domain (one wave per CMUX):
// res = c0 + trgsw [x] (c1 - c0), __CMUXNTT__ src/bootstrap_gpu.cu:197-285.  trgsw_ntt holds
// (k+1)l rows of two polynomials in the layout bk_to_ntt_kernel writes.
// ----------------------------------------------------------------------------------
__device__ __forceinline__ void cmux_wave(uint32_t* o, const double2* key, const uint32_t* p1, const uint32_t* p0,
                                          const WaveCtx& ctx, int lane)
{
    double A0[kRegs], A1[kRegs];
#pragma unroll
    for (int r = 0; r < kRegs; r++) { A0[r] = 0.0; A1[r] = 0.0; }
#pragma unroll 1
    for (int j = 0; j < 2; j++) {
        uint32_t temp[kRegs];
#pragma unroll
        for (int r = 0; r < kRegs; r++) {
            const int e = j * kN + lane + 64 * r;
            temp[r] = (p1[e] - p0[e] + decomp_offset()) ^ decomp_signmask();   // TRLWESubAndDecomposition :162-195
        }
#pragma unroll 1
        for (int d = 0; d < kL; d++) {
            const uint32_t pos = 32 - (d + 1) * kBgbit;
            double x[kRegs];
#pragma unroll
            for (int r = 0; r < kRegs; r++) x[r] = (double)(int32_t)__builtin_amdgcn_sbfe(temp[r], pos, (uint32_t)kBgbit);
            ntt_forward<true>(x, ctx);
            const double2* row = key + (size_t)(j * kL + d) * kN;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const double2 b0 = row[q * 64 + lane], b1 = row[512 + q * 64 + lane];
                A0[2 * q] += fpf::mulmod_wide(x[2 * q], b0.x);
                A0[2 * q + 1] += fpf::mulmod_wide(x[2 * q + 1], b0.y);
                A1[2 * q] += fpf::mulmod_wide(x[2 * q], b1.x);
                A1[2 * q + 1] += fpf::mulmod_wide(x[2 * q + 1], b1.y);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kRegs; r++) { A0[r] = fpf::reduce(A0[r]); A1[r] = fpf::reduce(A1[r]); }
    // c0 is read in full before the first word of `res` is written: res may be c0 (or c1) itself
    uint32_t base[2 * kRegs];
#pragma unroll
    for (int r = 0; r < kRegs; r++) { base[r] = p0[lane + 64 * r]; base[kRegs + r] = p0[kN + lane + 64 * r]; }
    ntt_inverse(A0, ctx);
    ntt_inverse(A1, ctx);
#pragma unroll
    for (int r = 0; r < kRegs; r++) {
        o[lane + 64 * r] = base[r] + fpf::lift_u32(A0[r]);
        o[kN + lane + 64 * r] = base[kRegs + r] + fpf::lift_u32(A1[r]);
    }
}

__global__ __launch_bounds__(kNttThreads) void cmux_kernel(
    uint32_t* res, const double* __restrict__ trgsw_ntt, const uint32_t* c1, const uint32_t* c0, int count,
    const NttTables* __restrict__ gt)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    load_tables_to_lds((double*)smem, gt);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * kNttWavesPerBlock + wave;
    if (g >= count) return;
    const WaveCtx ctx = make_wave_ctx(smem, kLdsTableBytes + wave * kTileBytes, 0, gt, lane);
    cmux_wave(res + (size_t)g * 2 * kN, (const double2*)(trgsw_ntt + (size_t)g * kBkStepDoubles), c1 + (size_t)g * 2 * kN,
              c0 + (size_t)g * 2 * kN, ctx, lane);
}

// the same on per-operation pointers: what the stream scheduler launches for the CMUXNTT calls of one dependence level
struct CmuxDesc {
    const uint32_t* c1;
    const uint32_t* c0;
    uint32_t* res;
    const double* trgsw_ntt;
};
__global__ __launch_bounds__(kNttThreads) void cmux_desc_kernel(const CmuxDesc* __restrict__ descs, int count, const NttTables* __restrict__ gt)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    load_tables_to_lds((double*)smem, gt);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * kNttWavesPerBlock + wave;
    if (g >= count) return;
    const WaveCtx ctx = make_wave_ctx(smem, kLdsTableBytes + wave * kTileBytes, 0, gt, lane);
    const CmuxDesc d = descs[g];
    cmux_wave(d.res, (const double2*)d.trgsw_ntt, d.c1, d.c0, ctx, lane);
}

// out = ca*in0 + cb*in1 + (0,..,off) over `words` words; grid-stride over ciphertexts
__global__ __launch_bounds__(256) void lincomb_kernel(const LinDesc* __restrict__ descs, int count, int words)
{
    for (int g = blockIdx.x; g < count; g += gridDim.x) {
        const LinDesc d = descs[g];
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
            uint32_t v = (uint32_t)d.ca * d.in0[i] + (uint32_t)d.cb * d.in1[i];
            if (i == words - 1) v += d.off;
            d.out[i] = v;
        }
    }
}

// ----------------------------------------------------------------------------------
// res = a (signed small) * b (torus) negacyclic mod 2^32, one wave per product: the
// device-side mirror of test/test_polynomial_mult_1024.cu (ForwardNTT, PointwiseMultiply,
// InverseNTT kernels).  Also exposes the raw forward+inverse round trip.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kNttThreads) void polymul_kernel(
    uint32_t* __restrict__ res, const int32_t* __restrict__ a, const uint32_t* __restrict__ b,
    int count, const NttTables* __restrict__ gt, double n_inverse)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tabs = (double*)smem;
    load_tables_to_lds(tabs, gt);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * kNttWavesPerBlock + wave;
    if (g >= count) return;
    const WaveCtx ctx = make_wave_ctx(smem, kLdsTableBytes + wave * kTileBytes, 0, gt, lane);
    double x[kRegs], y[kRegs];
#pragma unroll
    for (int r = 0; r < kRegs; r++) {
        x[r] = (double)a[(size_t)g * kN + lane + 64 * r];
        y[r] = (double)(int32_t)b[(size_t)g * kN + lane + 64 * r];
    }
    ntt_forward<false>(x, ctx);
    ntt_forward<false>(y, ctx);
#pragma unroll
    for (int r = 0; r < kRegs; r++) {
        y[r] = fpf::reduce(fpf::mulmod_wide(y[r], n_inverse));
        x[r] = fpf::reduce(fpf::mulmod_wide(x[r], y[r]));
    }
    ntt_inverse(x, ctx);
#pragma unroll
    for (int r = 0; r < kRegs; r++) res[(size_t)g * kN + lane + 64 * r] = fpf::lift_u32(x[r]);
}

}  // namespace cufhe_amd
