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
#include "ntt_r4.h"
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
    // radix-4 form (q4 below): the product of a block's stage-a and first stage-b twiddle, u w (forward) / v w (inverse), per lam
    // and per lane; the wave-uniform block's sits in the spare slot 7 of tu_fwd / tu_inv.  Contiguous, in this order.
    double uwb_fwd[8], uwb_inv[8];
    double uwc_fwd[64], uwc_inv[64];
};
static_assert(sizeof(Ntt512Tables) == (16 + 1008 + 144) * 8, "Ntt512Tables: [tu 16 | tb, tc 1008 | radix-4 products 144] doubles");
constexpr int kLds512TableDoubles = 2 * 7 * 8 + 2 * 7 * 64;     // tb_fwd .. tc_inv, contiguous: 1008
constexpr int kLds512TableBytes = kLds512TableDoubles * 8;      // 8064 per half

// ---- radix-4 form of the 512-point transforms (kernels_lvl2q.hip.h and the inverse waves of kernels_ll.hip.h) ----------
// A three-stage block on the eight registers of a lane (strides 4, 2, 1; twiddles tw0 | tw1, tw2 = I tw1 | tw3..6) is one radix-4
// pass over stages (a, b) -- groups {r, r + 2, r + 4, r + 6}, r = 0, 1: x0 = x[r], coarse partner x[r + 4] (twiddle w = tw0), fine
// partner x[r + 2] (u = tw1), the product u w from the uw* fields of Ntt512Tables (the uniform one in the spare slot 7 of
// tu_fwd / tu_inv) -- followed (forward) or preceded (inverse) by the radix-2 stage c: 30 + 30 + 32 operations where three radix-2
// stages take 96 (ntt_r4.h: ct_bfly4 / gs_bfly4, the product by I in four operations).  The bound of every register is carried
// through the passes at compile time as in ntt_r4.h (maximum over lanes, units of p; a layout change makes all eight equal to the
// largest): each product picks mulmod or mulmod_wide from the bound of its input and a reduction is spent on a register only when
// its bound asks for one -- 10 registers per inverse transform where the radix-2 schedule swept all eight twice.
namespace q4 {
struct B8 { double v[8]; };
constexpr B8 uniform8(double b) { return B8{{b, b, b, b, b, b, b, b}}; }
constexpr double max8(const B8& b)
{
    double m = 0;
    for (int i = 0; i < 8; i++) m = b.v[i] > m ? b.v[i] : m;
    return m;
}
constexpr B8 ct_r4_bounds(const B8& in)
{
    B8 o{};
    for (int r = 0; r < 2; r++) {
        const double a = in.v[r], pf = r4::after_product(in.v[r + 2]), pc = r4::after_product(in.v[r + 4]), pcf = r4::after_product(in.v[r + 6]);
        const double sum = r4::checked(r4::checked(a + pc) + r4::checked(pf + pcf)), dif = r4::checked(r4::checked(a + pc) + fpf::AFTER_MUL_ROOT4);
        o.v[r] = sum; o.v[r + 2] = sum; o.v[r + 4] = dif; o.v[r + 6] = dif;
    }
    return o;
}
// stage c of a forward block; REDUCE_ADDEND: the pass-through register is reduced first (the last stage of the transform)
template <bool REDUCE_ADDEND>
constexpr B8 ct_c_bounds(const B8& in)
{
    B8 o{};
    for (int g = 0; g < 4; g++) {
        const double t = r4::after_product(in.v[2 * g + 1]);
        o.v[2 * g] = o.v[2 * g + 1] = r4::checked((REDUCE_ADDEND ? r4::kReduced : in.v[2 * g]) + t);
    }
    return o;
}
constexpr B8 gs_c_bounds(const B8& in)
{
    B8 o{};
    for (int g = 0; g < 4; g++) {
        const double s = r4::checked(in.v[2 * g] + in.v[2 * g + 1]);
        o.v[2 * g] = s;
        o.v[2 * g + 1] = r4::after_product(s);
    }
    return o;
}
constexpr B8 gs_r4_bounds(const B8& in)
{
    B8 o{};
    for (int r = 0; r < 2; r++) {
        const double s0 = r4::checked(in.v[r] + in.v[r + 2]), s1 = r4::checked(in.v[r + 4] + in.v[r + 6]);
        o.v[r] = r4::checked(s0 + s1);
        o.v[r + 4] = r4::after_product(r4::checked(s0 + s1));
        o.v[r + 2] = r4::after_product(r4::checked(s0 + fpf::AFTER_MUL_ROOT4));
        o.v[r + 6] = r4::after_product(r4::checked(s0 + fpf::AFTER_MUL_ROOT4));
    }
    return o;
}
constexpr B8 reduce8_bounds(const B8& in, double limit)
{
    B8 o = in;
    for (int r = 0; r < 8; r++)
        if (in.v[r] > limit) o.v[r] = r4::kReduced;
    return o;
}
// schedules as types (S::in() = the bounds on entry)
template <int MICRO> struct U8 { static constexpr B8 in() { return uniform8(MICRO * 1e-6); } };
template <class S> struct CtR4 { static constexpr B8 in() { return ct_r4_bounds(S::in()); } };
template <class S, bool RA> struct CtC { static constexpr B8 in() { return ct_c_bounds<RA>(S::in()); } };
template <class S> struct GsC { static constexpr B8 in() { return gs_c_bounds(S::in()); } };
template <class S> struct GsR4 { static constexpr B8 in() { return gs_r4_bounds(S::in()); } };
template <class S> struct Xpose8 { static constexpr B8 in() { return uniform8(max8(S::in())); } };
template <class S, int LIMIT_MILLI> struct Red8 { static constexpr B8 in() { return reduce8_bounds(S::in(), LIMIT_MILLI * 0.001); } };

template <class S, int R = 0, int LIMIT_MILLI = 0>
__device__ __forceinline__ void reduce_above8(double (&x)[kRegs8])
{
    if constexpr (R < 8) {
        if constexpr (S::in().v[R] > LIMIT_MILLI * 0.001) x[R] = fpf::reduce(x[R]);
        reduce_above8<S, R + 1, LIMIT_MILLI>(x);
    }
}
// the radix-4 pass of a forward block on a lane with input schedule S
template <class S>
__device__ __forceinline__ void ct_r4_pass(double (&x)[kRegs8], double w, double u, double uw)
{
    constexpr B8 b = S::in();
    r4::ct_bfly4<r4::needs_wide(b.v[4]), r4::needs_wide(b.v[2]), r4::needs_wide(b.v[6])>(x[0], x[2], x[4], x[6], w, u, uw);
    r4::ct_bfly4<r4::needs_wide(b.v[5]), r4::needs_wide(b.v[3]), r4::needs_wide(b.v[7])>(x[1], x[3], x[5], x[7], w, u, uw);
}
template <class S, bool REDUCE_ADDEND, int G = 0, class TW>
__device__ __forceinline__ void ct_c_stage(double (&x)[kRegs8], const TW& tw)
{
    if constexpr (G < 4) {
        if constexpr (REDUCE_ADDEND) x[2 * G] = fpf::reduce(x[2 * G]);
        const double t = r4::product<r4::needs_wide(S::in().v[2 * G + 1])>(x[2 * G + 1], tw(3 + G));
        const double a = x[2 * G];
        x[2 * G] = a + t;
        x[2 * G + 1] = a - t;
        ct_c_stage<S, REDUCE_ADDEND, G + 1>(x, tw);
    }
}
template <class S, int G = 0, class TW>
__device__ __forceinline__ void gs_c_stage(double (&x)[kRegs8], const TW& tw)
{
    if constexpr (G < 4) {
        const double a = x[2 * G], b = x[2 * G + 1];
        x[2 * G] = a + b;
        x[2 * G + 1] = r4::product<r4::needs_wide(S::in().v[2 * G] + S::in().v[2 * G + 1])>(a - b, tw(3 + G));
        gs_c_stage<S, G + 1>(x, tw);
    }
}
template <class S>
__device__ __forceinline__ void gs_r4_pass(double (&x)[kRegs8], double w, double v, double vw)
{
    constexpr B8 b = S::in();
    {
        constexpr double s0 = b.v[0] + b.v[2], s1 = b.v[4] + b.v[6];
        r4::gs_bfly4<r4::needs_wide(s0 + s1), r4::needs_wide(s0 + fpf::AFTER_MUL_ROOT4), r4::needs_wide(s0 + fpf::AFTER_MUL_ROOT4)>(x[0], x[2], x[4], x[6], w, v, vw);
    }
    {
        constexpr double s0 = b.v[1] + b.v[3], s1 = b.v[5] + b.v[7];
        r4::gs_bfly4<r4::needs_wide(s0 + s1), r4::needs_wide(s0 + fpf::AFTER_MUL_ROOT4), r4::needs_wide(s0 + fpf::AFTER_MUL_ROOT4)>(x[1], x[3], x[5], x[7], w, v, vw);
    }
}
}  // namespace q4

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
// The inverse half transform in radix-4 form (q4 above), twiddles held by the caller as for ntt512_inverse_pinned EXCEPT that slot 2
// of each block holds the product v w (tu[2] = tu_inv[7], twb[2] = uwb_inv[lam], twc[2] = uwc_inv[lane]): no register more.  Out in
// layout A with |out| <= 1.28 p (what the tail's cheap lift of u0 + u1 accepts), 13 register reductions where the radix-2 schedule
// sweeps all eight twice.
namespace q4 {
constexpr int kLlLim = 2570, kLlOutLim = 1280;
using LC0 = U8<500100>;
using LC1 = Red8<GsC<LC0>, kLlLim>;
using LC2 = Red8<GsR4<LC1>, kLlLim>;
using LB0 = Xpose8<LC2>;
using LB1 = Red8<GsC<LB0>, kLlLim>;
using LB2 = Red8<GsR4<LB1>, kLlLim>;
using LA0 = Xpose8<LB2>;
using LA1 = Red8<GsC<LA0>, kLlLim>;
using LOut = Red8<GsR4<LA1>, kLlOutLim>;
static_assert(max8(LOut::in()) <= kLlOutLim * 0.001 && 2 * max8(LOut::in()) < 2.5705, "radix-4 inverse half transform: u0 + u1 must stay below 2^51 for lift_u32_small");
}  // namespace q4
__device__ __forceinline__ void ntt512_inverse_r4_pinned(double (&x)[kRegs8], const Wave512Ctx& c, const double (&tu)[7],
                                                         const double (&twb)[7], const double (&twc)[7])
{
    q4::gs_c_stage<q4::LC0>(x, TwArr{twc});                    // s8
    q4::reduce_above8<q4::GsC<q4::LC0>, 0, q4::kLlLim>(x);
    q4::gs_r4_pass<q4::LC1>(x, twc[0], twc[1], twc[2]);        // s7 s6
    q4::reduce_above8<q4::GsR4<q4::LC1>, 0, q4::kLlLim>(x);
    CUFHE_AMD_XPOSE8(c.c2, 8 * 72, c.b2, 64)          // C -> B
    q4::gs_c_stage<q4::LB0>(x, TwArr{twb});                    // s5
    q4::reduce_above8<q4::GsC<q4::LB0>, 0, q4::kLlLim>(x);
    q4::gs_r4_pass<q4::LB1>(x, twb[0], twb[1], twb[2]);        // s4 s3
    q4::reduce_above8<q4::GsR4<q4::LB1>, 0, q4::kLlLim>(x);
    CUFHE_AMD_XPOSE8(c.b1, 64, c.a1, 8 * 68)          // B -> A
    q4::gs_c_stage<q4::LA0>(x, TwArr{tu});                     // s2
    q4::reduce_above8<q4::GsC<q4::LA0>, 0, q4::kLlLim>(x);
    q4::gs_r4_pass<q4::LA1>(x, tu[0], tu[1], tu[2]);           // s1 s0
    q4::reduce_above8<q4::GsR4<q4::LA1>, 0, q4::kLlOutLim>(x);
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
