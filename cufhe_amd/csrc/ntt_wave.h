// ntt_wave.h -- one negacyclic NTT-1024 per wavefront, 16 coefficients per lane in VGPRs.
//
// Replaces the reference's shared-memory radix-2 NTT (SmallForwardNTT_1024 /
// SmallInverseNTT_1024, include/ntt_gpu/ntt_gpuntt.cuh:232-276,342-392: 512 threads,
// 5-6 block barriers per transform).  Same transform (merged-psi Cooley-Tukey forward,
// Gentleman-Sande inverse, twiddle index m+g), different machine mapping:
//
//   element index e (10 bits); the wave keeps three register layouts
//     A: lane = e[5:0]                 reg = e[9:6]      (natural: global/torus side)
//     B: lane = e[9:6] | e[1:0] << 4   reg = e[5:2]
//     C: lane = e[9:6] | e[5:4] << 4   reg = e[3:0]      (spectrum side: BK layout)
//   forward: stages 0-3 in A (strides 512..64 are register strides 8..1, twiddles are
//   wave-uniform scalars), A->B, stages 4-7 in B (15 per-lane twiddles), B->C, stages 8-9
//   in C.  Inverse mirrors it.  No block barrier anywhere.
//
//   A <-> B goes through a wave-private LDS tile.  Tile addressing is additive (slot =
//   pitch * e[9:6] + e[5:0]) so that the address is (per-lane base VGPR) + (compile-time
//   offset): no VALU instruction at all.  Pitch 66 makes the 32-lane ds_read_b64 of layout B
//   conflict-free (2*lambda + g) for A->B, pitch 65 does the same for the layout-B write of
//   B->A; the other side of each is the natural order.
//   B <-> C only exchanges lane bits 5:4 with two register bits: 32 v_permlane32/16_swap
//   instructions, no LDS (measured 1.4 % faster than a second LDS round trip).
//
// Two forms of the butterflies live here: the radix-2 stages below (key conversion, the workgroup-per-rotation kernels, the
// N = 2048 half transforms) and, at the end of the file, the radix-4 passes of ntt_r4.h that blind_rotate_kernel and the
// wave-per-rotation kernel of the parameter sets run (same layouts and layout changes; bounds per register at compile time).
//
// Lazy-reduction schedule of the radix-2 form (bounds in units of p, see fpfield.h; checked on the host by
// tests/host/host_model.cpp through tests/test_fpfield.py):
//   forward, gadget digits in (|x| <= 32): stages 0 and 1 are one exact radix-4 butterfly on the inputs (roots I, zeta and
//            zeta^3 I = -zeta: values stay below 2^42.2 = 0.006 p, nothing is reduced); then
//            after s2..s9: .51 1.06 1.66 2.32 3.05 3.85 4.72 6.18
//            (only stage 9 needs mulmod_wide: kept wide as before, its input is now 4.72)
//   forward, 32-bit words in (key conversion): every stage reduces; .5 1.05 ... 5.67 7.22 8.92
//            with stages 8 and 9 wide
//   inverse  in <= .5         s9 1.0  s8 2.0  s7 4.0  s6 8.0(wide) reduce  s5 1.0  s4 2.0
//            s3 4.0  s2 8.0(wide) reduce  s1 1.0  s0 2.0
//   pointwise: digits spectrum <= 6.18 -> each wide product <= 1.601, six of them 9.61 < 10.285
#pragma once
#include <hip/hip_runtime.h>
#include "fpfield.h"
#include "ntt_r4.h"

namespace cufhe_amd {

constexpr int kN = 1024;
constexpr int kRegs = 16;            // coefficients per lane
constexpr int kTbCount = 15;         // per-lane twiddles of stages 4-7
constexpr int kTcCount = 12;         // per-lane twiddles of stages 8-9
constexpr int kTbpStride = 18;       // doubles per lambda in the packed stage 4-7 table (15 used)
constexpr int kTcpStride = 14;       // doubles per lane in the packed stage 8-9 table (12 used)
constexpr int kTileSlots = 66 * 16;  // 8-byte slots in a wave's transpose tile
constexpr int kTileBytes = kTileSlots * 8;   // 8448

// The lazy-reduction schedule of the header comment, replayed at compile time (units of p).
// Bound on the spectrum of a polynomial whose coefficients are at most `digit_max` in
// magnitude, after the ten forward stages of ntt_forward<SMALL_IN = true>; -1 if a stage's
// input would exceed what its multiplication accepts.
constexpr double forward_digit_spectrum_bound(double digit_max)
{
    // stages 0 and 1 are one exact radix-4 butterfly on the inputs (ct_four_stages): every partial sum must stay an exact double
    const double s1 = digit_max * (1.0 + fpf::ROOT4 + fpf::ROOT8 + fpf::ROOT8 * fpf::ROOT8 * fpf::ROOT8);
    if (s1 >= 9007199254740992.0 / 64.0) return -1.0;         // every partial sum of the butterfly is below s1: 6 bits of headroom below 2^53 (Bg = 2^10: 2^46.2)
    double b = s1 / fpf::P;                                   // 2^42.2 / p = 0.006 for Bg = 2^6 (0.5 until the second group of stage 1 was made exact)
    for (int s = 2; s <= 8; s++) {                            // stages 2..8: mulmod
        if (b >= fpf::LIM_NARROW) return -1.0;
        b = b + fpf::after_mulmod(b);
    }
    if (b >= fpf::LIM_WIDE) return -1.0;                      // stage 9: mulmod_wide
    return b + fpf::after_mulmod_wide(b);
}
// The same for ntt_forward<SMALL_IN = false> (any 32-bit words in: every stage reduces, stages 8 and 9 wide): 8.92
constexpr double forward_words_spectrum_bound()
{
    double b = 4294967296.0 / fpf::P;
    for (int s = 0; s <= 7; s++) {
        if (b >= fpf::LIM_NARROW) return -1.0;
        b = b + fpf::after_mulmod(b);
    }
    for (int s = 8; s <= 9; s++) {
        if (b >= fpf::LIM_WIDE) return -1.0;
        b = b + fpf::after_mulmod_wide(b);
    }
    return b;
}
static_assert(forward_words_spectrum_bound() > 0 && forward_words_spectrum_bound() < fpf::LIM_WIDE, "forward NTT schedule (32-bit words in)");
// `rows` wide products of a spectrum bounded by s accumulate without reduction
constexpr bool pointwise_sum_fits(double s, int rows)
{
    return s > 0 && s < fpf::LIM_WIDE && rows * fpf::after_mulmod_wide(s) < fpf::LIM_WIDE;
}
// inverse transform (ntt_inverse): input reduced to <= 0.5 (+ a tie); two Gentleman-Sande stages,
// then twice the four stages of gs_four_stages (narrow, wide, reduce, narrow, narrow).  A GS
// butterfly doubles the bound of its sum output and multiplies a difference of twice the input.
constexpr double inverse_output_bound()
{
    double b = 0.5001;
    for (int s = 0; s < 2; s++) {                 // stages 9, 8: mulmod(u - v)
        if (2 * b >= fpf::LIM_NARROW) return -1.0;
        b = 2 * b;
    }
    for (int half = 0; half < 2; half++) {        // stages 7..4, then 3..0
        if (2 * b >= fpf::LIM_NARROW) return -1.0;    // narrow
        b = 2 * b;
        if (2 * b >= fpf::LIM_WIDE) return -1.0;      // wide
        b = 2 * b;                                    // <= 8: reduce() accepts anything below 2^53
        b = 0.5001;
        for (int s = 0; s < 2; s++) {                 // two narrow stages
            if (2 * b >= fpf::LIM_NARROW) return -1.0;
            b = 2 * b;
        }
    }
    return b;                                         // 2.0: below the 2^51 / p = 2.57 lift_u32_small accepts
}
constexpr bool inverse_schedule_fits() { return inverse_output_bound() > 0 && inverse_output_bound() < 2.57; }
static_assert(inverse_schedule_fits(), "inverse NTT lazy-reduction schedule exceeds the FP64 mantissa");

// twiddle tables, generated on the host with exact integer arithmetic (capi.cpp)
struct NttTables {
    double tu_fwd[16];               // [k] k<15: root[2^lvl + j], lvl=floor(log2(k+1)), j=k+1-2^lvl
    double tu_inv[16];
    double tb_fwd[kTbCount * 16];    // [k][lambda]: root[16*2^lvl + lambda*2^lvl + j]
    double tb_inv[kTbCount * 16];
    double tc_fwd[kTcCount * 64];    // [k][lane]: k<4: root[256 + (lambda<<4|h<<2|k)]; else root[512 + (lambda<<5|h<<3|(k-4))]
    double tc_inv[kTcCount * 64];
    // The same per-lane twiddles PACKED per lane (filled for the r4 tables only): a lane's fifteen stage 4-7 twiddles and its
    // twelve stage 8-9 twiddles are contiguous, so a transform fetches them with 8 + 6 ds_read_b128 instead of 27 ds_read_b64
    // (the compiler pairs those into ds_read2_b64, which moves 128 B per LDS clock where ds_read_b128 moves 256).  The strides
    // -- 18 doubles = 144 bytes per lambda, 14 doubles = 112 bytes per lane -- put the sixteen lanes of every ds_read_b128 lane
    // group on sixteen different 4-bank slots.  [tbp_fwd | tbp_inv | tcp_fwd | tcp_inv] is one contiguous block.
    double tbp_fwd[16 * kTbpStride];
    double tbp_inv[16 * kTbpStride];
    double tcp_fwd[64 * kTcpStride];
    double tcp_inv[64 * kTcpStride];
};
constexpr int kLdsTablePackedDoubles = 2 * 16 * kTbpStride + 2 * 64 * kTcpStride;   // 2368
constexpr int kLdsTablePackedBytes = kLdsTablePackedDoubles * 8;                     // 18944
constexpr int kLdsTableDoubles = 2 * kTbCount * 16 + 2 * kTcCount * 64;   // 2016
constexpr int kLdsTableBytes = kLdsTableDoubles * 8;                        // 16128

// cooperative copy global -> LDS (whole workgroup), caller barriers afterwards
__device__ __forceinline__ void load_tables_to_lds(double* lds, const NttTables* g)
{
    const double* src = g->tb_fwd;   // tb_fwd, tb_inv, tc_fwd, tc_inv are contiguous
    for (int i = threadIdx.x; i < kLdsTableDoubles; i += blockDim.x) lds[i] = src[i];
}

// the packed tables (NttTables::tbp_fwd ..) instead: kLdsTablePackedBytes at `lds`
__device__ __forceinline__ void load_packed_tables_to_lds(double* lds, const NttTables* g)
{
    const double* src = g->tbp_fwd;
    for (int i = threadIdx.x; i < kLdsTablePackedDoubles; i += blockDim.x) lds[i] = src[i];
}

// Per-lane LDS byte addresses, computed once per kernel and kept in VGPRs.
struct WaveCtx {
    char* a65;   // layout A, pitch 65:  tile + 8*lane                 (+ 8*65*r)
    char* a66;   // layout A, pitch 66:  same base                      (+ 8*66*r)
    char* b65;   // layout B, pitch 65:  tile + 8*(65*lambda + g)       (+ 32*r)
    char* b66;   // layout B, pitch 66:  tile + 8*(66*lambda + g)       (+ 32*r)
    const char* tb_fwd;   // LDS tables + 8*lambda  (+ 128*k)
    const char* tb_inv;
    const char* tc_fwd;   // LDS tables + 8*lane    (+ 512*k)
    const char* tc_inv;
    const NttTables* gt;  // global tables (uniform scalars)
    // optional: [tu_fwd[16] | tu_inv[16]] in LDS (256 bytes).  Read through the scalar cache at their use (TwUniform) the
    // stage 0-3 twiddles cost an s_load followed at once by s_waitcnt lgkmcnt(0) -- the scalar-load latency exposed AND the
    // wave's LDS reads drained -- in every transform; as broadcast LDS reads issued ahead of the transpose in front of
    // their phase (TU_LDS variants below) they cost nothing.  Used where waves run in lock-step (kernels_lvl2.hip.h); in
    // the wave-per-rotation kernel the partner wave covers the stall (measured: 38.63 ms either way).
    const char* tu_l;
};

// LDS-DMA of one 1 KiB piece (16 bytes per lane): global `src` (per lane) -> LDS `dst` (wave-uniform base; lane L lands at
// dst + 16 L).  Written as inline assembly ON PURPOSE: issued through __builtin_amdgcn_global_load_lds the compiler knows that
// an asynchronous LDS write is in flight, cannot tell which LDS reads it may alias (every per-lane LDS base here is opaque) and
// puts `s_waitcnt vmcnt(0)` in front of the NEXT ds_read whatever it reads -- the wave then sits out the whole L2 round trip
// of the piece it has just requested, once per key row (found in the ISA: in front of the first ds_read_b128 of the
// pointwise product).  The protocol of the callers never reads a buffer before the barrier that follows the wait below, so no
// such wait is needed; with the request invisible to the compiler none is emitted.  lds_dma_wait_all() is the wait the
// callers place in front of that barrier.  M0 carries the LDS base (a reserved register the compiler sets before each of its own uses).
__device__ __forceinline__ void lds_dma16(const void* src, char* dst_uniform)
{
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)dst_uniform;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(base) : "memory", "m0");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// A byte offset the compiler cannot see through: `lds + opaque(off)` keeps the full per-lane
// address in ONE VGPR, so every access is "VGPR + small immediate".  Without it hipcc folds
// the wave's tile base into the 16-bit DS offset field and, once that base exceeds 64 KiB,
// spends a v_add per access to rebuild the address.
__device__ __forceinline__ int opaque(int off)
{
    asm volatile("" : "+v"(off));
    return off;
}

// lds: start of the workgroup's dynamic LDS; tile_off / tables_off: byte offsets in it
__device__ __forceinline__ WaveCtx make_wave_ctx(char* lds, int tile_off, int tables_off,
                                                 const NttTables* gt, int lane)
{
    const int lam = lane & 15, hi = lane >> 4;
    WaveCtx c;
    c.a65 = lds + opaque(tile_off + 8 * lane);
    c.a66 = c.a65;
    c.b65 = lds + opaque(tile_off + 8 * (65 * lam + hi));
    c.b66 = lds + opaque(tile_off + 8 * (66 * lam + hi));
    c.tb_fwd = lds + opaque(tables_off + 8 * lam);
    c.tb_inv = c.tb_fwd + 8 * (kTbCount * 16);
    c.tc_fwd = lds + opaque(tables_off + 8 * (2 * kTbCount * 16) + 8 * lane);
    c.tc_inv = c.tc_fwd + 8 * (kTcCount * 64);
    c.gt = gt;
    c.tu_l = nullptr;
    return c;
}

// WaveCtx whose tb_* / tc_* point into the PACKED tables (load_packed_tables_to_lds): tb_fwd + 16 i is the double2 of
// twiddles 2i, 2i + 1 of this lane's lambda, tc_fwd + 16 i that of its stage 8-9 twiddles
__device__ __forceinline__ WaveCtx make_wave_ctx_packed(char* lds, int tile_off, int tables_off, const NttTables* gt, int lane)
{
    WaveCtx c = make_wave_ctx(lds, tile_off, tables_off, gt, lane);
    c.tb_fwd = lds + opaque(tables_off + 8 * kTbpStride * (lane & 15));
    c.tb_inv = c.tb_fwd + 8 * 16 * kTbpStride;
    c.tc_fwd = lds + opaque(tables_off + 8 * 2 * 16 * kTbpStride + 8 * kTcpStride * lane);
    c.tc_inv = c.tc_fwd + 8 * 64 * kTcpStride;
    return c;
}
__device__ __forceinline__ double2 lds_ld2(const char* p, int off) { return *(const double2*)(p + off); }
// COUNT packed twiddles of this lane into registers (ds_read_b128 each two).  (No CUFHE_AMD_ABL_NO_TW form: with constants in
// place of these twiddles hipcc folds a fifth of the transform's arithmetic away -- 470 FP64 instructions less in the kernel --
// so that "ablation" measures a different program; its -13 % turned out to be exactly that: profiles/r04_radix4_and_latency.md.)
template <int COUNT>
__device__ __forceinline__ void load_packed(double (&tw)[COUNT], const char* base)
{
#pragma unroll
    for (int i = 0; i < (COUNT + 1) / 2; i++) {
        const double2 v = lds_ld2(base, 16 * i);
        tw[2 * i] = v.x;
        if (2 * i + 1 < COUNT) tw[2 * i + 1] = v.y;
    }
}
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_NO_TW)
__device__ __forceinline__ double lds_ld(const char* p, int off) { return 12345678.0 + (double)off + (double)(uintptr_t)p * 1e-30; }
#else
__device__ __forceinline__ double lds_ld(const char* p, int off) { return *(const double*)(p + off); }
#endif
__device__ __forceinline__ void lds_st(char* p, int off, double v) { *(double*)(p + off) = v; }

// ---- butterflies -----------------------------------------------------------------
template <bool WIDE>
__device__ __forceinline__ void ct_bfly(double& a, double& b, double w)
{
    const double t = WIDE ? fpf::mulmod_wide(b, w) : fpf::mulmod(b, w);
    const double u = a;
    a = u + t;
    b = u - t;
}
template <bool WIDE>
__device__ __forceinline__ void gs_bfly(double& a, double& b, double w)
{
    const double u = a, v = b;
    a = u + v;
    b = WIDE ? fpf::mulmod_wide(u - v, w) : fpf::mulmod(u - v, w);
}

struct TwUniform {                   // stages 0-3: wave-uniform scalars from global memory
    const double* t;
    __device__ __forceinline__ double operator()(int k) const { return t[k]; }
};
struct TwLane {                      // stages 4-7: tb[k][lambda] from LDS
    const char* t;
    __device__ __forceinline__ double operator()(int k) const { return lds_ld(t, 128 * k); }
};

// Four radix-2 stages on register strides 8,4,2,1 with twiddles tw(0..14)
// (tw(0) | tw(1..2) | tw(3..6) | tw(7..14)); identical in layouts A and B.
// multiplication by a root so small that the product is exact: no reduction
__device__ __forceinline__ void ct_bfly_exact(double& a, double& b, double w)
{
    const double u = a, v = b;
    a = __builtin_fma(v, w, u);       // exact: an integer below 2^53
    b = __builtin_fma(-v, w, u);
}

// Stages 0 and 1 of a polynomial of small integers (gadget digits) as one exact radix-4 butterfly, see ct_four_stages<SMALL_IN>
__device__ __forceinline__ void ct_exact_first_two(double (&x)[kRegs])
{
    constexpr double kZ3 = fpf::ROOT8 * fpf::ROOT8 * fpf::ROOT8;          // zeta^3 = 160 989 184 000, exact
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double a = x[r], a1 = x[r + 4], b = x[r + 8], b1 = x[r + 12];
        const double u = __builtin_fma(b, fpf::ROOT4, a), v = __builtin_fma(-b, fpf::ROOT4, a);
        const double u1 = __builtin_fma(b1, fpf::ROOT4, a1);
        const double t = __builtin_fma(a1, kZ3, b1 * fpf::ROOT8);
        x[r] = __builtin_fma(u1, fpf::ROOT8, u);
        x[r + 4] = __builtin_fma(-u1, fpf::ROOT8, u);
        x[r + 8] = v + t;
        x[r + 12] = v - t;
    }
}

// SMALL_IN: |x| <= 32 on entry and the twiddles are those of stages 0-3 (tw(0) = I,
// tw(1) = zeta): stage 0 and the first group of stage 1 use ct_bfly_exact.
// EXACT0 (with SMALL_IN false): stage 0 multiplies by a root so small -- tw(0) = zeta, 13 bits, against inputs below
// 2^33 -- that the product is exact (the first stage of half 0 of the N = 2048 transform)
template <bool SMALL_IN, class TW, bool EXACT0 = false>
__device__ __forceinline__ void ct_four_stages(double (&x)[kRegs], const TW& tw)
{
    if (SMALL_IN) {
        // Stages 0 and 1 as one exact radix-4 butterfly on the four ORIGINAL inputs a, a', b, b' (elements e, e + 256, e + 512, e + 768):
        //   u = a + I b, u' = a' + I b', v = a - I b                 stage 0 (I = zeta^2, 25 bits)
        //   x[r] = u + zeta u', x[r + 4] = u - zeta u'                stage 1, first group (zeta, 13 bits)
        //   x[r + 8], x[r + 12] = v +- zeta^3 (a' - I b')             stage 1, second group: its twiddle zeta^3 has 37 bits, but
        //                       = v +- (zeta^3 a' + zeta b')          zeta^3 I = zeta^5 = -zeta, so it only ever meets an input, not a product
        // Nine exact FP64 operations per four elements (fourteen with the second group through a modular product); every
        // value stays below |x| (1 + I + zeta + zeta^3) = 2^42.2 for gadget digits.
        ct_exact_first_two(x);
    } else {
        const double w0 = tw(0);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (EXACT0) ct_bfly_exact(x[r], x[r + 8], w0);
            else ct_bfly<false>(x[r], x[r + 8], w0);
        }
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const double w = tw(1 + g);
#pragma unroll
            for (int r = 0; r < 4; r++) ct_bfly<false>(x[8 * g + r], x[8 * g + r + 4], w);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const double w = tw(3 + g);
#pragma unroll
        for (int r = 0; r < 2; r++) ct_bfly<false>(x[4 * g + r], x[4 * g + r + 2], w);
    }
#pragma unroll
    for (int g = 0; g < 8; g++) ct_bfly<false>(x[2 * g], x[2 * g + 1], tw(7 + g));
}
template <class TW>
__device__ __forceinline__ void gs_four_stages(double (&x)[kRegs], const TW& tw)
{
#pragma unroll
    for (int g = 0; g < 8; g++) gs_bfly<false>(x[2 * g], x[2 * g + 1], tw(7 + g));
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const double w = tw(3 + g);
#pragma unroll
        for (int r = 0; r < 2; r++) gs_bfly<true>(x[4 * g + r], x[4 * g + r + 2], w);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = fpf::reduce(x[r]);
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const double w = tw(1 + g);
#pragma unroll
        for (int r = 0; r < 4; r++) gs_bfly<false>(x[8 * g + r], x[8 * g + r + 4], w);
    }
    {
        const double w = tw(0);
#pragma unroll
        for (int r = 0; r < 8; r++) gs_bfly<false>(x[r], x[r + 8], w);
    }
}

// One Gentleman-Sande stage with register stride S (1, 2, 4, 8) and twiddles tw(first + g), and the four stages of
// gs_four_stages applied to TWO polynomials stage by stage (two inverse transforms in one wave: their dependent chains
// and LDS round trips cover each other when the partner wave on the SIMD is parked)
template <int S, bool WIDE, class TW>
__device__ __forceinline__ void gs_stage(double (&x)[kRegs], const TW& tw, int first)
{
#pragma unroll
    for (int g = 0; g < 8 / S; g++) {
        const double w = tw(first + g);
#pragma unroll
        for (int r = 0; r < S; r++) gs_bfly<WIDE>(x[2 * S * g + r], x[2 * S * g + r + S], w);
    }
}
template <class TW>
__device__ __forceinline__ void gs_four_stages2(double (&x)[kRegs], double (&y)[kRegs], const TW& tw)
{
    gs_stage<1, false>(x, tw, 7);
    gs_stage<1, false>(y, tw, 7);
    gs_stage<2, true>(x, tw, 3);
    gs_stage<2, true>(y, tw, 3);
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = fpf::reduce(x[r]);
#pragma unroll
    for (int r = 0; r < 16; r++) y[r] = fpf::reduce(y[r]);
    gs_stage<4, false>(x, tw, 1);
    gs_stage<4, false>(y, tw, 1);
    gs_stage<8, false>(x, tw, 0);
    gs_stage<8, false>(y, tw, 0);
}

// ---- layout changes through the wave-private LDS tile -----------------------------
// DS operations of one wave execute in issue order, so a wave may reuse its own tile
// without any barrier.
// Ablation switches (timing-only diagnostic builds, results are wrong when set):
//   CUFHE_AMD_ABL_NO_XPOSE  skip the LDS transposes     CUFHE_AMD_ABL_NO_TW  constant twiddles
//   CUFHE_AMD_ABL_NO_BK     (kernels.hip.h) no bootstrapping-key loads
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_NO_XPOSE)
#define CUFHE_AMD_XPOSE(WBASE, WSTRIDE, RBASE, RSTRIDE) { asm volatile("" ::: "memory"); }
#else
// The compiler fence keeps the tile accesses of consecutive layout changes in program
// order: the per-lane base pointers are opaque on purpose, so hipcc cannot see that they address
// the same tile and may otherwise hoist the stores of an independent later transform above the
// loads of an earlier one (observed with two back-to-back 512-point transforms).
#define CUFHE_AMD_XPOSE(WBASE, WSTRIDE, RBASE, RSTRIDE)                              \
    {                                                                                \
        asm volatile("" ::: "memory");                                               \
        _Pragma("unroll") for (int r = 0; r < kRegs; r++) lds_st(WBASE, (WSTRIDE) * r, x[r]); \
        _Pragma("unroll") for (int r = 0; r < kRegs; r++) x[r] = lds_ld(RBASE, (RSTRIDE) * r); \
    }
#endif

// The same two layout changes through a HALF-SIZE tile (8 rows of pitch 66: 4224 bytes) in two passes of eight
// registers: pass p moves the elements with e[9] = p.  A -> B: all lanes store registers 8p .. 8p + 7 as rows 0 .. 7, then
// the lanes whose layout-B row e[9:6] has e[9] = p (lane bit 3) read their sixteen registers; B -> A: those lanes store
// their sixteen registers, then all lanes read registers 8p .. 8p + 7.  Twice the DS instructions on one side (half of
// them with half the lanes), no VALU instruction more; for kernels whose LDS is the scarce resource (kernels_lvl2.hip.h).
// WaveCtx for it: make_wave_ctx_half (b65 / b66 built from e[8:6]).
constexpr int kHalfTileBytes = 66 * 8 * 8;   // 4224
template <bool A_TO_B>
__device__ __forceinline__ void xpose_half_tile(double (&x)[kRegs], char* abase, char* bbase)
{
    const bool upper = (threadIdx.x & 8) != 0;      // e[9] of this lane's layout-B row
    double t[8];                                     // eight registers beside x, not sixteen
    if (A_TO_B) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 8; r++) lds_st(abase, 8 * 66 * r, x[8 * p + r]);
            asm volatile("" ::: "memory");
            if (upper == (p == 1)) {                 // registers 0-7 of x have been stored by now (pass 0), t is free
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] = lds_ld(bbase, 32 * k);
#pragma unroll
                for (int k = 0; k < 8; k++) t[k] = lds_ld(bbase, 32 * (k + 8));
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < 8; k++) x[8 + k] = t[k];
    } else {
        asm volatile("" ::: "memory");
        if (!upper) {
#pragma unroll
            for (int k = 0; k < kRegs; k++) lds_st(bbase, 32 * k, x[k]);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 8; r++) t[r] = lds_ld(abase, 8 * 65 * r);
        asm volatile("" ::: "memory");
        if (upper) {
#pragma unroll
            for (int k = 0; k < kRegs; k++) lds_st(bbase, 32 * k, x[k]);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 8; r++) x[8 + r] = lds_ld(abase, 8 * 65 * r);      // x is dead in every lane by now
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 8; r++) x[r] = t[r];
    }
}

// ---- layout change B <-> C without LDS -------------------------------------------------
// B: lane = lambda | g << 4, reg = 4 h + m      C: lane = lambda | h << 4, reg = 4 m + g
// (g = e[1:0], h = e[5:4], m = e[3:2]).  For every m this is a 4 x 4 transpose between the
// wave's four 16-lane rows and four registers: v_permlane32_swap exchanges lane bit 5 with
// register bit h1, v_permlane16_swap lane bit 4 with register bit h0.  32 VALU instructions,
// no LDS round trip.  The same sequence maps C back to B.
__device__ __forceinline__ void swap_halves32(double& lo_reg, double& hi_reg)
{
    // lo_reg's upper 32 lanes <-> hi_reg's lower 32 lanes
    unsigned a0 = (unsigned)__double2loint(lo_reg), a1 = (unsigned)__double2hiint(lo_reg);
    unsigned b0 = (unsigned)__double2loint(hi_reg), b1 = (unsigned)__double2hiint(hi_reg);
    auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    lo_reg = __hiloint2double((int)r1[0], (int)r0[0]);
    hi_reg = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ void swap_rows16(double& lo_reg, double& hi_reg)
{
    // lo_reg's odd 16-lane rows <-> hi_reg's even rows
    unsigned a0 = (unsigned)__double2loint(lo_reg), a1 = (unsigned)__double2hiint(lo_reg);
    unsigned b0 = (unsigned)__double2loint(hi_reg), b1 = (unsigned)__double2hiint(hi_reg);
    auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
    lo_reg = __hiloint2double((int)r1[0], (int)r0[0]);
    hi_reg = __hiloint2double((int)r1[1], (int)r0[1]);
}
// x indexed by the source layout's register number; returns in the destination's numbering
__device__ __forceinline__ void xpose_bc_permlane(double (&x)[kRegs])
{
    // source reg = 4 a + m with a = (a1 a0) the bits to move into the lane index
#pragma unroll
    for (int m = 0; m < 4; m++) {
#pragma unroll
        for (int a0 = 0; a0 < 2; a0++) swap_halves32(x[4 * a0 + m], x[4 * (2 + a0) + m]);      // a1: 0 <-> 1
#pragma unroll
        for (int a1 = 0; a1 < 2; a1++) swap_rows16(x[4 * (2 * a1) + m], x[4 * (2 * a1 + 1) + m]);   // a0: 0 <-> 1
    }
    // now reg 4 b + m holds the element whose old lane-row index was b: rename to 4 m + b
    double y[kRegs];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int m = 0; m < 4; m++) y[4 * m + b] = x[4 * b + m];
#pragma unroll
    for (int r = 0; r < kRegs; r++) x[r] = y[r];
}

// C -> B: the bits to move into the lane index are the LOW register bits (reg = 4 m + g)
__device__ __forceinline__ void xpose_cb_permlane(double (&x)[kRegs])
{
#pragma unroll
    for (int m = 0; m < 4; m++) {
#pragma unroll
        for (int g0 = 0; g0 < 2; g0++) swap_halves32(x[4 * m + g0], x[4 * m + 2 + g0]);          // g1: 0 <-> 1
#pragma unroll
        for (int g1 = 0; g1 < 2; g1++) swap_rows16(x[4 * m + 2 * g1], x[4 * m + 2 * g1 + 1]);    // g0: 0 <-> 1
    }
    // reg 4 m + b now holds the element whose old lane-row index was b (= h): rename to 4 b + m
    double y[kRegs];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int m = 0; m < 4; m++) y[4 * b + m] = x[4 * m + b];
#pragma unroll
    for (int r = 0; r < kRegs; r++) x[r] = y[r];
}

struct TwArr {                       // twiddles already in registers
    const double* t;
    __device__ __forceinline__ double operator()(int k) const { return t[k]; }
};

// The per-lane twiddles of a phase are fetched into registers BEFORE the transpose that
// precedes the phase: their LDS latency then overlaps the transpose instead of being paid
// butterfly by butterfly (measured: twiddle fetches placed at their use cost 38 % of the
// kernel, profiles/r01_ablation.md).

// forward, phase A: stages 0-3 in layout A (natural order in)
template <bool SMALL_IN>
__device__ __forceinline__ void ntt_forward_a(double (&x)[kRegs], const WaveCtx& c)
{
    ct_four_stages<SMALL_IN>(x, TwUniform{c.gt->tu_fwd});
}
// the same with the fifteen stage 0-3 twiddles in registers: load_tu(tu, c.tu_l) / load_tu(tu, c.tu_l + 128) for the
// inverse, placed by the caller far enough ahead of the transform
__device__ __forceinline__ void load_tu(double (&tu)[15], const char* tu_lds)
{
#pragma unroll
    for (int k = 0; k < 15; k++) tu[k] = lds_ld(tu_lds, 8 * k);
}
template <bool SMALL_IN, bool EXACT0 = false>
__device__ __forceinline__ void ntt_forward_a_tu(double (&x)[kRegs], const double (&tu)[15])
{
    ct_four_stages<SMALL_IN, TwArr, EXACT0>(x, TwArr{tu});
}
// forward, phases B and C: out in layout C (spectrum order).  WIDE8: stage 8 inputs may
// exceed 5.142 p (true for 32-bit inputs, not for gadget digits).
template <bool WIDE8, bool HALF_TILE = false>
__device__ __forceinline__ void ntt_forward_bc(double (&x)[kRegs], const WaveCtx& c)
{
    double twb[kTbCount];
#pragma unroll
    for (int k = 0; k < kTbCount; k++) twb[k] = lds_ld(c.tb_fwd, 128 * k);
    if (HALF_TILE) xpose_half_tile<true>(x, c.a66, c.b66);
    else CUFHE_AMD_XPOSE(c.a66, 8 * 66, c.b66, 32)        // A -> B
    ct_four_stages<false>(x, TwArr{twb});
    double twc[kTcCount];
#pragma unroll
    for (int k = 0; k < kTcCount; k++) twc[k] = lds_ld(c.tc_fwd, 512 * k);
    xpose_bc_permlane(x);                            // B -> C in registers
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
        for (int r = 0; r < 2; r++) ct_bfly<WIDE8>(x[4 * g + r], x[4 * g + r + 2], twc[g]);
    }
#pragma unroll
    for (int g = 0; g < 8; g++) ct_bfly<true>(x[2 * g], x[2 * g + 1], twc[4 + g]);
}
// SMALL_IN: the input is a gadget-digit polynomial (|x| <= 32); otherwise 32-bit words
template <bool SMALL_IN, bool HALF_TILE = false>
__device__ __forceinline__ void ntt_forward(double (&x)[kRegs], const WaveCtx& c)
{
    ntt_forward_a<SMALL_IN>(x, c);
    ntt_forward_bc<!SMALL_IN, HALF_TILE>(x, c);
}

// inverse: x in layout C with |x| <= p/2, out in layout A with |x| <= 2p, NOT scaled by
// 1/N (N^-1 is folded into the NTT-domain bootstrapping key)
// twc: the twelve stage 9-8 twiddles of this lane (c.tc_inv + 512 k), fetched by the caller -- ahead of a barrier, say
template <bool HALF_TILE = false, bool TU_LDS = false>
__device__ __forceinline__ void ntt_inverse_twc(double (&x)[kRegs], const WaveCtx& c, const double (&twc)[kTcCount])
{
#pragma unroll
    for (int g = 0; g < 8; g++) gs_bfly<false>(x[2 * g], x[2 * g + 1], twc[4 + g]);
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
        for (int r = 0; r < 2; r++) gs_bfly<false>(x[4 * g + r], x[4 * g + r + 2], twc[g]);
    }
    double twb[kTbCount];
#pragma unroll
    for (int k = 0; k < kTbCount; k++) twb[k] = lds_ld(c.tb_inv, 128 * k);
    xpose_cb_permlane(x);                            // C -> B in registers
    gs_four_stages(x, TwArr{twb});
    double tu[15];
    if (TU_LDS) load_tu(tu, c.tu_l + 128);           // in flight during the transpose
    if (HALF_TILE) xpose_half_tile<false>(x, c.a65, c.b65);
    else CUFHE_AMD_XPOSE(c.b65, 32, c.a65, 8 * 65)        // B -> A
    if (TU_LDS) gs_four_stages(x, TwArr{tu});
    else gs_four_stages(x, TwUniform{c.gt->tu_inv});
}
// Two inverse transforms with the SAME tables (two sums of one half) in one wave, stage by stage; the twiddles are loaded
// once.  The tile is used by x's transpose, then by y's.
template <bool HALF_TILE = false, bool TU_LDS = false>
__device__ __forceinline__ void ntt_inverse2_twc(double (&x)[kRegs], double (&y)[kRegs], const WaveCtx& c, const double (&twc)[kTcCount])
{
    gs_stage<1, false>(x, TwArr{twc}, 4);
    gs_stage<1, false>(y, TwArr{twc}, 4);
    gs_stage<2, false>(x, TwArr{twc}, 0);
    gs_stage<2, false>(y, TwArr{twc}, 0);
    double twb[kTbCount];
#pragma unroll
    for (int k = 0; k < kTbCount; k++) twb[k] = lds_ld(c.tb_inv, 128 * k);
    xpose_cb_permlane(x);                            // C -> B in registers
    xpose_cb_permlane(y);
    gs_four_stages2(x, y, TwArr{twb});
    double tu[15];
    if (TU_LDS) load_tu(tu, c.tu_l + 128);
    if (HALF_TILE) { xpose_half_tile<false>(x, c.a65, c.b65); xpose_half_tile<false>(y, c.a65, c.b65); }
    else { CUFHE_AMD_XPOSE(c.b65, 32, c.a65, 8 * 65) { double (&x)[kRegs] = y; CUFHE_AMD_XPOSE(c.b65, 32, c.a65, 8 * 65) } }
    if (TU_LDS) gs_four_stages2(x, y, TwArr{tu});
    else gs_four_stages2(x, y, TwUniform{c.gt->tu_inv});
}
template <bool HALF_TILE = false, bool TU_LDS = false>
__device__ __forceinline__ void ntt_inverse(double (&x)[kRegs], const WaveCtx& c)
{
    double twc[kTcCount];
#pragma unroll
    for (int k = 0; k < kTcCount; k++) twc[k] = lds_ld(c.tc_inv, 512 * k);
    ntt_inverse_twc<HALF_TILE, TU_LDS>(x, c, twc);
}


// ==================================================================================================================
// Radix-4 transforms (ntt_r4.h) on the r4 twiddle tables (capi.hip: fill_tables(..., r4 = true)) in their PACKED form
// (WaveCtx from make_wave_ctx_packed, LDS copy by load_packed_tables_to_lds): the same layouts,
// layout changes and twiddle fetch placement as above, 30 instead of 32 operations per four elements and two stages,
// and reductions only on the registers whose compile-time bound asks for one.
// ==================================================================================================================

// STORE: every group of the last pass is written to the transpose tile (A side, pitch 66) as soon as it is computed, so the
// LDS writes of the layout change run beside the rest of the pass instead of after it; ntt_forward_digits_bc_r4<.., STORED = true>
// then only reads.  (The tile's earlier readers -- rotate_sub, the previous transform -- produced the data these values are
// computed from, so the data dependence orders the stores behind them; the compiler fence in front keeps them behind any
// independent tile access of an earlier transform, as CUFHE_AMD_XPOSE does.)
// The twelve wave-uniform twiddles of stages 2-3 (r4 table slots 3..14), loaded ONCE per kernel and pinned in scalar registers:
// read through the scalar cache in every transform they cost an s_load followed by s_waitcnt lgkmcnt(0), which also drains
// the wave's LDS queue -- with the tile stores of STORE below in flight that wait sits in the middle of the pass.
struct TuFwdPinned {
    double t[12];
    __device__ __forceinline__ void load(const NttTables* gt)
    {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            double v = gt->tu_fwd[3 + k];
            asm volatile("" : "+s"(v));       // opaque: kept in an SGPR pair, never re-loaded
            t[k] = v;
        }
    }
    __device__ __forceinline__ double operator()(int k) const { return t[k - 3]; }
};
template <int DIGIT_MAX, bool STORE = false, class TW = TwUniform>
__device__ __forceinline__ void ntt_forward_digits_a_r4(double (&x)[kRegs], const WaveCtx& c, const TW* pinned = nullptr, double (*twb_ahead)[kTbCount] = nullptr)
{
    using A0 = typename r4::FwdDigits<DIGIT_MAX>::A0;
    ct_exact_first_two(x);
    if (!STORE) {
        r4::ct_pass_lo<A0, 3, 7>(x, TwUniform{c.gt->tu_fwd});
    } else {
        const TW& tw = *pinned;
        if (twb_ahead) load_packed(*twb_ahead, c.tb_fwd);      // the stage 4-7 twiddles of ntt_forward_digits_bc_r4, requested a pass ahead of their use
        asm volatile("" ::: "memory");
        r4::Group<A0, false, 0>::ct(x, tw(3), tw(7), tw(8));
#pragma unroll
        for (int r = 0; r < 4; r++) lds_st(c.a66, 8 * 66 * r, x[r]);
        r4::Group<A0, false, 1>::ct(x, tw(4), tw(9), tw(10));
#pragma unroll
        for (int r = 4; r < 8; r++) lds_st(c.a66, 8 * 66 * r, x[r]);
        r4::Group<A0, false, 2>::ct(x, tw(5), tw(11), tw(12));
#pragma unroll
        for (int r = 8; r < 12; r++) lds_st(c.a66, 8 * 66 * r, x[r]);
        r4::Group<A0, false, 3>::ct(x, tw(6), tw(13), tw(14));
#pragma unroll
        for (int r = 12; r < 16; r++) lds_st(c.a66, 8 * 66 * r, x[r]);
    }
}
template <int DIGIT_MAX, bool HALF_TILE = false, bool STORED = false>
__device__ __forceinline__ void ntt_forward_digits_bc_r4(double (&x)[kRegs], const WaveCtx& c, const double* twb_loaded = nullptr)
{
    using F = r4::FwdDigits<DIGIT_MAX>;
    static_assert(r4::valid(F::Spectrum::in()), "radix-4 forward transform of gadget digits: a value exceeds 2^53");
    static_assert(!(HALF_TILE && STORED), "the half-size tile takes the layout change in two passes");
    double twb[kTbCount];
    if (twb_loaded) {
#pragma unroll
        for (int k = 0; k < kTbCount; k++) twb[k] = twb_loaded[k];
    } else {
        load_packed(twb, c.tb_fwd);
    }
    if (HALF_TILE) xpose_half_tile<true>(x, c.a66, c.b66);
    else if (STORED) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < kRegs; r++) x[r] = lds_ld(c.b66, 32 * r);
    }
    else CUFHE_AMD_XPOSE(c.a66, 8 * 66, c.b66, 32)        // A -> B
    r4::ct_pass_hi<typename F::B0>(x, TwArr{twb});
    r4::ct_pass_lo<typename F::B1, 3, 7>(x, TwArr{twb});
    double twc[kTcCount];
    load_packed(twc, c.tc_fwd);
    xpose_bc_permlane(x);                            // B -> C in registers
    r4::reduce_mask<F::kReduceC>(x);
    r4::ct_pass_lo<typename F::C1, 0, 4>(x, TwArr{twc});
}

template <class S0, bool HALF_TILE, bool TWB_LOADED>
__device__ __forceinline__ void ntt_inverse_r4_tw(double (&x)[kRegs], const WaveCtx& c, const double (&twc)[kTcCount], double (&twb)[kTbCount]);
template <class S0, bool HALF_TILE = false>
__device__ __forceinline__ void ntt_inverse_r4(double (&x)[kRegs], const WaveCtx& c)
{
    double twc[kTcCount], twb[kTbCount];
    load_packed(twc, c.tc_inv);
    ntt_inverse_r4_tw<S0, HALF_TILE, false>(x, c, twc, twb);
}
// The same with the twelve stage 9-8 twiddles already in registers (a caller with several inverse transforms in a row loads
// them once, ahead of the work in front of the first); TWB_LOADED: the fifteen stage 7-4 twiddles as well, else they are
// fetched here (into twb, which a following transform may then re-use)
template <class S0, bool HALF_TILE, bool TWB_LOADED>
__device__ __forceinline__ void ntt_inverse_r4_tw(double (&x)[kRegs], const WaveCtx& c, const double (&twc)[kTcCount], double (&twb)[kTbCount])
{
    using V = r4::Inverse<S0>;
    static_assert(r4::valid(V::Out::in()), "radix-4 inverse transform: a value exceeds 2^53");
    r4::gs_pass_lo<S0, 0, 4>(x, TwArr{twc});
    r4::reduce_above<typename V::C1, V::kLimit>(x);
    if (!TWB_LOADED) load_packed(twb, c.tb_inv);
    xpose_cb_permlane(x);                            // C -> B in registers
    r4::gs_pass_lo<typename V::B0, 3, 7>(x, TwArr{twb});
    r4::reduce_above<r4::AfterGs<typename V::B0, false>, V::kLimit>(x);
    if (HALF_TILE) {
        r4::gs_pass_hi<typename V::B1>(x, TwArr{twb});
        r4::reduce_above<r4::AfterGs<typename V::B1, true>, V::kLimit>(x);
        xpose_half_tile<false>(x, c.a65, c.b65);
    } else {
        // B -> A with every group of the last pass stored as soon as it is computed (cf. ntt_forward_digits_a_r4<.., STORE>)
        using B1o = r4::AfterGs<typename V::B1, true>;
        const double w = twb[0], v = twb[1], vw = twb[2];
        double tu[15];                                   // stage 3-0 twiddles (wave-uniform): their scalar loads go out ahead of the last pass
#pragma unroll                                           // (as LDS broadcasts instead: 36.01 against 35.95 ms on one box)
        for (int k = 0; k < 15; k++) tu[k] = c.gt->tu_inv[k];
        asm volatile("" ::: "memory");
        r4::Group<typename V::B1, true, 0>::gs(x, w, v, vw);
        r4::reduce_above_group<B1o, V::kLimit, 0>(x);
#pragma unroll
        for (int k = 0; k < 16; k += 4) lds_st(c.b65, 32 * k, x[k]);
        r4::Group<typename V::B1, true, 1>::gs(x, w, v, vw);
        r4::reduce_above_group<B1o, V::kLimit, 1>(x);
#pragma unroll
        for (int k = 1; k < 16; k += 4) lds_st(c.b65, 32 * k, x[k]);
        r4::Group<typename V::B1, true, 2>::gs(x, w, v, vw);
        r4::reduce_above_group<B1o, V::kLimit, 2>(x);
#pragma unroll
        for (int k = 2; k < 16; k += 4) lds_st(c.b65, 32 * k, x[k]);
        r4::Group<typename V::B1, true, 3>::gs(x, w, v, vw);
        r4::reduce_above_group<B1o, V::kLimit, 3>(x);
#pragma unroll
        for (int k = 3; k < 16; k += 4) lds_st(c.b65, 32 * k, x[k]);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < kRegs; r++) x[r] = lds_ld(c.a65, 8 * 65 * r);
        r4::gs_pass_lo<typename V::A0, 3, 7>(x, TwArr{tu});
        r4::reduce_above<r4::AfterGs<typename V::A0, false>, V::kLimit>(x);
        r4::gs_pass_hi<typename V::A1>(x, TwArr{tu});
        return;
    }
    r4::gs_pass_lo<typename V::A0, 3, 7>(x, TwUniform{c.gt->tu_inv});
    r4::reduce_above<r4::AfterGs<typename V::A0, false>, V::kLimit>(x);
    r4::gs_pass_hi<typename V::A1>(x, TwUniform{c.gt->tu_inv});
}
// acc[r] += torus word of x[r] (the centred lift), the cheaper lift where the register's bound allows it
template <class OUT, int R = 0>
__device__ __forceinline__ void lift_add(uint32_t (&acc)[kRegs], const double (&x)[kRegs])
{
    if constexpr (R < kRegs) {
        if constexpr (OUT::in().v[R] < 2.57) acc[R] += fpf::lift_u32_small(x[R]);
        else acc[R] += fpf::lift_u32(x[R]);
        lift_add<OUT, R + 1>(acc, x);
    }
}

}  // namespace cufhe_amd
