// ntt_wave512.h -- one HALF of the 1024-point negacyclic NTT per wavefront.
//
// X^1024 + 1 = (X^512 - I)(X^512 + I), I = psi^512: after the first Cooley-Tukey stage
// u_h[e] = a[e] +- I a[e + 512] (h = 0, 1) the transform splits into two independent
// 512-point transforms whose twiddles are the h-th half of every later stage of the full
// table, root_h[m + g] = root[2m + h m + g].  The low-latency blind-rotate kernel gives each
// half to its own wave: half the dependent instruction chain per wave, twice the waves.
//
// 8 coefficients per lane; e = 64 lam + 8 kap + c (three octal digits).  Layouts:
//   A: lane = 8 kap + c,  reg = lam   (natural order: e = lane + 64 reg)
//   B: lane = lam + 8 c,  reg = kap
//   C: lane = lam + 8 kap, reg = c    (spectrum side)
// Forward: stages 0-2 in A (7 wave-uniform twiddles), A->B, stages 3-5 in B (7 twiddles per
// lam), B->C, stages 6-8 in C (7 twiddles per lane).  Both layout changes go through the
// wave's LDS tile with two different slot maps, each free of bank conflicts on both sides:
//   A<->B: slot = 68 lam + 8 kap + c        B<->C: slot = lam + 8 kap + 72 c
// The spectrum value of position p = 512 h + e of the full transform (the order the NTT-domain
// bootstrapping key is stored in) ends up in wave h, lane lam + 8 kap, register c.
//
// Lazy-reduction schedule (units of p, cf. ntt_wave.h): forward, inputs far below p:
//   .5 1.05 1.65 2.31 3.04 3.83 4.71 5.66 7.22 after stages 0..8 (only stage 8 is wide);
// inverse, inputs <= .5: s8 1  s7 2  s6 4  s5 8 (wide) reduce  s4 1  s3 2  s2 4  s1 8 (wide)
//   reduce  s0 1.
#pragma once
#include "ntt_wave.h"

namespace cufhe_amd {

constexpr int kH = 512;               // points of a half transform
constexpr int kRegs8 = 8;
constexpr int kTile512Slots = 568;    // max slot of either map + 1
constexpr int kTile512Bytes = kTile512Slots * 8;   // 4544

// tables of ONE half transform (host: capi.hip build_tables_512)
struct Ntt512Tables {
    double tu_fwd[8];                 // [k] k<7: root_h[2^lvl + j], lvl = floor(log2(k+1)), j = k+1-2^lvl
    double tu_inv[8];
    double tb_fwd[7 * 8];             // [k][lam]: root_h[8*2^lvl + lam*2^lvl + j]
    double tb_inv[7 * 8];
    double tc_fwd[7 * 64];            // [k][lane]: root_h[64*2^lvl + mu*2^lvl + j], mu = 8 lam + kap
    double tc_inv[7 * 64];
};
constexpr int kLds512TableDoubles = 2 * 7 * 8 + 2 * 7 * 64;     // tb_fwd .. tc_inv, contiguous: 1008
constexpr int kLds512TableBytes = kLds512TableDoubles * 8;      // 8064 per half

struct Wave512Ctx {
    char* a1;        // A side of the A<->B map: tile + 8 lane                     (+ 8*68 reg)
    char* b1;        // B side of the A<->B map: tile + 8 (68 lam + c)             (+ 64 reg)
    char* b2;        // B side of the B<->C map: tile + 8 (lam + 72 c)             (+ 64 reg)
    char* c2;        // C side of the B<->C map: tile + 8 lane                     (+ 8*72 reg)
    const char* tb_fwd;   // LDS tables + 8 lam    (+ 64 k)
    const char* tb_inv;
    const char* tc_fwd;   // LDS tables + 8 lane   (+ 512 k)
    const char* tc_inv;
    const Ntt512Tables* gt;
};

// tables_off: byte offset in LDS of this half's [tb_fwd | tb_inv | tc_fwd | tc_inv] copy
__device__ __forceinline__ Wave512Ctx make_wave512_ctx(char* lds, int tile_off, int tables_off,
                                                       const Ntt512Tables* gt, int lane)
{
    const int lo = lane & 7, hi = lane >> 3;
    Wave512Ctx c;
    c.a1 = lds + opaque(tile_off + 8 * lane);
    c.b1 = lds + opaque(tile_off + 8 * (68 * lo + hi));
    c.b2 = lds + opaque(tile_off + 8 * (lo + 72 * hi));
    c.c2 = c.a1;
    c.tb_fwd = lds + opaque(tables_off + 8 * lo);
    c.tb_inv = c.tb_fwd + 8 * (7 * 8);
    c.tc_fwd = lds + opaque(tables_off + 8 * (2 * 7 * 8) + 8 * lane);
    c.tc_inv = c.tc_fwd + 8 * (7 * 64);
    c.gt = gt;
    return c;
}

// three radix-2 stages on register strides 4, 2, 1 with twiddles tw(0) | tw(1..2) | tw(3..6)
template <bool WIDE_LAST, class TW>
__device__ __forceinline__ void ct_three_stages(double (&x)[kRegs8], const TW& tw)
{
    const double w0 = tw(0);
#pragma unroll
    for (int r = 0; r < 4; r++) ct_bfly<false>(x[r], x[r + 4], w0);
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const double w = tw(1 + g);
#pragma unroll
        for (int r = 0; r < 2; r++) ct_bfly<false>(x[4 * g + r], x[4 * g + r + 2], w);
    }
#pragma unroll
    for (int g = 0; g < 4; g++) ct_bfly<WIDE_LAST>(x[2 * g], x[2 * g + 1], tw(3 + g));
}
// inverse order; the stage with index WIDE (0 = stride 1, 1 = stride 2, 2 = stride 4; -1: none)
// takes inputs up to 4 p (wide multiply) and is followed by a reduction of all registers
template <int WIDE, class TW>
__device__ __forceinline__ void gs_three_stages(double (&x)[kRegs8], const TW& tw)
{
#pragma unroll
    for (int g = 0; g < 4; g++) gs_bfly<WIDE == 0>(x[2 * g], x[2 * g + 1], tw(3 + g));
    if (WIDE == 0) {
#pragma unroll
        for (int r = 0; r < kRegs8; r++) x[r] = fpf::reduce(x[r]);
    }
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const double w = tw(1 + g);
#pragma unroll
        for (int r = 0; r < 2; r++) gs_bfly<WIDE == 1>(x[4 * g + r], x[4 * g + r + 2], w);
    }
    if (WIDE == 1) {
#pragma unroll
        for (int r = 0; r < kRegs8; r++) x[r] = fpf::reduce(x[r]);
    }
    {
        const double w = tw(0);
#pragma unroll
        for (int r = 0; r < 4; r++) gs_bfly<WIDE == 2>(x[r], x[r + 4], w);
    }
}

// The compiler fence keeps the tile accesses of consecutive layout changes in program order
// (the hardware executes one wave's DS operations in order, but hipcc is free to hoist the
// stores of an independent later transform above the loads of an earlier one: it does not see
// that the per-lane base pointers, made opaque on purpose, address the same tile).
#define CUFHE_AMD_XPOSE8(WBASE, WSTRIDE, RBASE, RSTRIDE)                                \
    {                                                                                   \
        asm volatile("" ::: "memory");                                                  \
        _Pragma("unroll") for (int r = 0; r < kRegs8; r++) lds_st(WBASE, (WSTRIDE) * r, x[r]); \
        _Pragma("unroll") for (int r = 0; r < kRegs8; r++) x[r] = lds_ld(RBASE, (RSTRIDE) * r); \
    }

// forward half transform: x = u_h in layout A (|x| far below p), out in layout C, |out| <= 7.22 p.
// tu: the seven stage 0-2 twiddles held by the caller (a wave whose role is fixed loads them once per kernel instead of an
// s_load followed by a full lgkmcnt drain in every transform)
__device__ __forceinline__ void ntt512_forward_tu(double (&x)[kRegs8], const Wave512Ctx& c, const double (&tu)[7])
{
    ct_three_stages<false>(x, TwArr{tu});
    double twb[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twb[k] = lds_ld(c.tb_fwd, 64 * k);
    CUFHE_AMD_XPOSE8(c.a1, 8 * 68, c.b1, 64)          // A -> B
    ct_three_stages<false>(x, TwArr{twb});
    double twc[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twc[k] = lds_ld(c.tc_fwd, 512 * k);
    CUFHE_AMD_XPOSE8(c.b2, 64, c.c2, 8 * 72)          // B -> C
    ct_three_stages<true>(x, TwArr{twc});
}
// The same with stage 0 already applied by the caller (the low-latency kernels fold it, exactly, into the split of the gadget
// digits: kernels_ll.hip.h, ll_split_first_stages): stages 1 and 2 with tu[1..6], then as above.
__device__ __forceinline__ void ntt512_forward_tu_from1(double (&x)[kRegs8], const Wave512Ctx& c, const double (&tu)[7])
{
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const double w = tu[1 + g];
#pragma unroll
        for (int r = 0; r < 2; r++) ct_bfly<false>(x[4 * g + r], x[4 * g + r + 2], w);
    }
#pragma unroll
    for (int g = 0; g < 4; g++) ct_bfly<false>(x[2 * g], x[2 * g + 1], tu[3 + g]);
    double twb[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twb[k] = lds_ld(c.tb_fwd, 64 * k);
    CUFHE_AMD_XPOSE8(c.a1, 8 * 68, c.b1, 64)          // A -> B
    ct_three_stages<false>(x, TwArr{twb});
    double twc[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twc[k] = lds_ld(c.tc_fwd, 512 * k);
    CUFHE_AMD_XPOSE8(c.b2, 64, c.c2, 8 * 72)          // B -> C
    ct_three_stages<true>(x, TwArr{twc});
}
__device__ __forceinline__ void ntt512_inverse_tu(double (&x)[kRegs8], const Wave512Ctx& c, const double (&tu)[7])
{
    double twc[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twc[k] = lds_ld(c.tc_inv, 512 * k);
    double twb[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twb[k] = lds_ld(c.tb_inv, 64 * k);
    gs_three_stages<-1>(x, TwArr{twc});               // s8 s7 s6: .5 -> 4
    CUFHE_AMD_XPOSE8(c.c2, 8 * 72, c.b2, 64)          // C -> B
    gs_three_stages<0>(x, TwArr{twb});                // s5 (wide, reduce) s4 s3: -> 2
    CUFHE_AMD_XPOSE8(c.b1, 64, c.a1, 8 * 68)          // B -> A
    gs_three_stages<1>(x, TwArr{tu});                 // s2 s1 (wide, reduce) s0: -> 1
}
// The same two transforms with ALL of the wave's twiddles held by the caller: a wave of the low-latency kernels keeps one role (row
// or inverse) for the whole kernel, so the seven per-lam and seven per-lane twiddles of its direction are fetched once, not once per
// CMux step (an inverse wave is alone on its SIMD: the fetch sat at the head of its dependent chain).
__device__ __forceinline__ void ntt512_forward_pinned_from1(double (&x)[kRegs8], const Wave512Ctx& c, const double (&tu)[7],
                                                            const double (&twb)[7], const double (&twc)[7])
{
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const double w = tu[1 + g];
#pragma unroll
        for (int r = 0; r < 2; r++) ct_bfly<false>(x[4 * g + r], x[4 * g + r + 2], w);
    }
#pragma unroll
    for (int g = 0; g < 4; g++) ct_bfly<false>(x[2 * g], x[2 * g + 1], tu[3 + g]);
    CUFHE_AMD_XPOSE8(c.a1, 8 * 68, c.b1, 64)          // A -> B
    ct_three_stages<false>(x, TwArr{twb});
    CUFHE_AMD_XPOSE8(c.b2, 64, c.c2, 8 * 72)          // B -> C
    ct_three_stages<true>(x, TwArr{twc});
}
__device__ __forceinline__ void ntt512_inverse_pinned(double (&x)[kRegs8], const Wave512Ctx& c, const double (&tu)[7],
                                                      const double (&twb)[7], const double (&twc)[7])
{
    gs_three_stages<-1>(x, TwArr{twc});               // s8 s7 s6: .5 -> 4
    CUFHE_AMD_XPOSE8(c.c2, 8 * 72, c.b2, 64)          // C -> B
    gs_three_stages<0>(x, TwArr{twb});                // s5 (wide, reduce) s4 s3: -> 2
    CUFHE_AMD_XPOSE8(c.b1, 64, c.a1, 8 * 68)          // B -> A
    gs_three_stages<1>(x, TwArr{tu});                 // s2 s1 (wide, reduce) s0: -> 1
}
__device__ __forceinline__ void ntt512_forward(double (&x)[kRegs8], const Wave512Ctx& c)
{
    ct_three_stages<false>(x, TwUniform{c.gt->tu_fwd});
    double twb[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twb[k] = lds_ld(c.tb_fwd, 64 * k);
    CUFHE_AMD_XPOSE8(c.a1, 8 * 68, c.b1, 64)          // A -> B
    ct_three_stages<false>(x, TwArr{twb});
    double twc[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twc[k] = lds_ld(c.tc_fwd, 512 * k);
    CUFHE_AMD_XPOSE8(c.b2, 64, c.c2, 8 * 72)          // B -> C
    ct_three_stages<true>(x, TwArr{twc});
}

// The STAND-ALONE 512-point negacyclic transform (tables512[2]: psi_1024 = psi_2048^2, so its first twiddles are root[1] = I,
// root[2] = zeta, root[3] = zeta^3) of a gadget-digit polynomial, |x| <= 32: stages 0 and 1 as one exact radix-4 butterfly on the
// inputs a, a', b, b' = elements e, e + 128, e + 256, e + 384 (ntt_wave.h: ct_four_stages<SMALL_IN>; zeta^3 I = -zeta), nine FP64
// operations per four elements instead of two general butterfly stages (sixteen); every value stays below 2^42.2.
__device__ __forceinline__ void ntt512_forward_small(double (&x)[kRegs8], const Wave512Ctx& c)
{
    constexpr double kZ3 = fpf::ROOT8 * fpf::ROOT8 * fpf::ROOT8;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const double a = x[r], a1 = x[r + 2], b = x[r + 4], b1 = x[r + 6];
        const double u = __builtin_fma(b, fpf::ROOT4, a), v = __builtin_fma(-b, fpf::ROOT4, a);
        const double u1 = __builtin_fma(b1, fpf::ROOT4, a1);
        const double t = __builtin_fma(a1, kZ3, b1 * fpf::ROOT8);
        x[r] = __builtin_fma(u1, fpf::ROOT8, u);
        x[r + 2] = __builtin_fma(-u1, fpf::ROOT8, u);
        x[r + 4] = v + t;
        x[r + 6] = v - t;
    }
#pragma unroll
    for (int g = 0; g < 4; g++) ct_bfly<false>(x[2 * g], x[2 * g + 1], c.gt->tu_fwd[3 + g]);
    double twb[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twb[k] = lds_ld(c.tb_fwd, 64 * k);
    CUFHE_AMD_XPOSE8(c.a1, 8 * 68, c.b1, 64)          // A -> B
    ct_three_stages<false>(x, TwArr{twb});
    double twc[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twc[k] = lds_ld(c.tc_fwd, 512 * k);
    CUFHE_AMD_XPOSE8(c.b2, 64, c.c2, 8 * 72)          // B -> C
    ct_three_stages<true>(x, TwArr{twc});
}

// inverse half transform: x in layout C with |x| <= p/2, out = u_h in layout A, |out| <= p,
// not scaled (N^-1 is folded into the bootstrapping key)
__device__ __forceinline__ void ntt512_inverse(double (&x)[kRegs8], const Wave512Ctx& c)
{
    double twc[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twc[k] = lds_ld(c.tc_inv, 512 * k);
    double twb[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twb[k] = lds_ld(c.tb_inv, 64 * k);
    gs_three_stages<-1>(x, TwArr{twc});               // s8 s7 s6: .5 -> 4
    CUFHE_AMD_XPOSE8(c.c2, 8 * 72, c.b2, 64)          // C -> B
    gs_three_stages<0>(x, TwArr{twb});                // s5 (wide, reduce) s4 s3: -> 2
    CUFHE_AMD_XPOSE8(c.b1, 64, c.a1, 8 * 68)          // B -> A
    gs_three_stages<1>(x, TwUniform{c.gt->tu_inv});   // s2 s1 (wide, reduce) s0: -> 1
}

}  // namespace cufhe_amd
