// ntt_r4.h -- radix-4 butterflies over the FP64 prime field and their lazy-reduction bounds at compile time.
//
// The reference's transform is radix-2: N/2 log N general modular products (include/ntt_gpu/ntt_gpuntt.cuh:170-224,
// CooleyTukeyUnit / GentlemanSandeUnit), and so are the transforms of ntt_wave.h / ntt_wave512.h (ct_bfly / gs_bfly, 8 FP64 operations per
// butterfly).  Two consecutive merged-psi stages on four elements use the twiddles w | u, I u (root[2m + 1] = root[2m] psi^(N/2)
// and psi^(N/2) = I, the fourth root of unity), so they can be written with THREE general products and one product by I:
//
//   Cooley-Tukey (forward):  A = x0, B = w x_c, C = u x_f, D = (u w) x_cf          (c: coarse bit set, f: fine bit set)
//                            x0' = (A + B) + (C + D)      x_f'  = (A + B) - (C + D)
//                            x_c' = (A - B) + I (C - D)   x_cf' = (A - B) - I (C - D)
//   Gentleman-Sande (inverse, twiddles v | -I v of the fine stage, w of the coarse one):
//                            P = x0 - x_f, Q = I (x_c - x_cf)
//                            x0' = (x0 + x_f) + (x_c + x_cf)       x_c'  = w ((x0 + x_f) - (x_c + x_cf))
//                            x_f' = v (P - Q)                      x_cf' = (v w) (P + Q)
//
// and over p = I^2 + 1 the product by I costs four operations (fpfield.h: mul_root4) against six: 30 operations per four
// elements and two stages instead of 32.  The tables keep their shape; the slot of the second fine twiddle (I u, -I v) holds
// the product u w (v w) instead (capi.hip: fill_tables with r4 = true).
//
// What radix 4 changes is the growth of the values between reductions: the pass-through element x0 collects three
// reduced addends per pass instead of one per stage.  The schedule is therefore no longer a chain of scalars
// (ntt_wave.h: forward_digit_spectrum_bound) but tracked PER REGISTER, at compile time: a `Sched` type carries the
// magnitude bound of each of the 16 registers of a lane (maximum over lanes, units of p); every pass picks mulmod or
// mulmod_wide per product from the bound of its input and yields the Sched of its outputs; layout changes take the
// maximum over the registers that meet in one; a reduction is spent only on the registers that need one (a full sweep
// costs 48 operations, the all-sums register of a pass 3).  A bound that exceeds what its use accepts poisons the
// schedule and fails the static_assert of the kernel that instantiates it.
//
// Host-callable: tests/host/host_model.cpp runs these very functions on 64 emulated lanes (worst-case inputs included).
#pragma once
#include "fpfield.h"

#if defined(__HIPCC__)
#define R4_HD __host__ __device__ __forceinline__
#else
#define R4_HD inline
#endif

namespace cufhe_amd {
namespace r4 {

struct RegBounds {
    double v[16];
};
constexpr double kPoison = 1.0e30;      // a violated bound; every later bound derived from it stays above any limit
constexpr double kReduced = 0.5001;     // |reduce(x)| (a tie adds 1)

constexpr RegBounds uniform(double b)
{
    RegBounds r{};
    for (int i = 0; i < 16; i++) r.v[i] = b;
    return r;
}
constexpr double max_of(const RegBounds& b)
{
    double m = 0;
    for (int i = 0; i < 16; i++) m = b.v[i] > m ? b.v[i] : m;
    return m;
}
constexpr bool valid(const RegBounds& b) { return max_of(b) < fpf::LIM_WIDE; }
// bound of a general product whose input is bounded by b: narrow below 2^52, wide below 2^53
constexpr bool needs_wide(double b) { return b >= fpf::LIM_NARROW; }
constexpr double after_product(double b)
{
    return b >= fpf::LIM_WIDE ? kPoison : needs_wide(b) ? fpf::after_mulmod_wide(b) : fpf::after_mulmod(b);
}
constexpr double checked(double b) { return b >= fpf::LIM_WIDE ? kPoison : b; }     // any value must stay below 2^53

// Register index bits of the two passes of a 16-register block: HI couples registers r, r + 4, r + 8, r + 12 (coarse
// stride 8, fine stride 4), LO couples 4g .. 4g + 3 (coarse stride 2, fine stride 1).
template <bool HI>
constexpr int reg_of(int group, int coarse, int fine)
{
    return HI ? group + 8 * coarse + 4 * fine : 4 * group + 2 * coarse + fine;
}

template <bool HI>
constexpr RegBounds ct_bounds(const RegBounds& in)
{
    RegBounds out{};
    for (int g = 0; g < 4; g++) {
        const double a = in.v[reg_of<HI>(g, 0, 0)];
        const double mb = after_product(in.v[reg_of<HI>(g, 1, 0)]);
        const double mc = after_product(in.v[reg_of<HI>(g, 0, 1)]);
        const double md = after_product(in.v[reg_of<HI>(g, 1, 1)]);
        const double sum = checked(checked(a + mb) + checked(mc + md));
        const double dif = checked(checked(a + mb) + fpf::AFTER_MUL_ROOT4);
        out.v[reg_of<HI>(g, 0, 0)] = sum;
        out.v[reg_of<HI>(g, 0, 1)] = sum;
        out.v[reg_of<HI>(g, 1, 0)] = dif;
        out.v[reg_of<HI>(g, 1, 1)] = dif;
    }
    return out;
}
template <bool HI>
constexpr RegBounds gs_bounds(const RegBounds& in)
{
    RegBounds out{};
    for (int g = 0; g < 4; g++) {
        const double s0 = checked(in.v[reg_of<HI>(g, 0, 0)] + in.v[reg_of<HI>(g, 0, 1)]);
        const double s1 = checked(in.v[reg_of<HI>(g, 1, 0)] + in.v[reg_of<HI>(g, 1, 1)]);
        out.v[reg_of<HI>(g, 0, 0)] = checked(s0 + s1);
        out.v[reg_of<HI>(g, 1, 0)] = after_product(checked(s0 + s1));
        out.v[reg_of<HI>(g, 0, 1)] = after_product(checked(s0 + fpf::AFTER_MUL_ROOT4));
        out.v[reg_of<HI>(g, 1, 1)] = after_product(checked(s0 + fpf::AFTER_MUL_ROOT4));
    }
    return out;
}
// Layout changes of ntt_wave.h.  A <-> B (through the LDS tile): all four register bits become lane bits.
constexpr RegBounds xpose_all_bounds(const RegBounds& in) { return uniform(max_of(in)); }
// B (reg = 4 h + m) -> C (reg = 4 m + g): h goes to the lanes, g comes from them
constexpr RegBounds xpose_bc_bounds(const RegBounds& in)
{
    RegBounds out{};
    for (int m = 0; m < 4; m++) {
        double mx = 0;
        for (int h = 0; h < 4; h++) mx = in.v[4 * h + m] > mx ? in.v[4 * h + m] : mx;
        for (int g = 0; g < 4; g++) out.v[4 * m + g] = mx;
    }
    return out;
}
// C (reg = 4 m + g) -> B (reg = 4 h + m)
constexpr RegBounds xpose_cb_bounds(const RegBounds& in)
{
    RegBounds out{};
    for (int m = 0; m < 4; m++) {
        double mx = 0;
        for (int g = 0; g < 4; g++) mx = in.v[4 * m + g] > mx ? in.v[4 * m + g] : mx;
        for (int h = 0; h < 4; h++) out.v[4 * h + m] = mx;
    }
    return out;
}
// registers above `limit` are reduced
constexpr RegBounds reduce_bounds(const RegBounds& in, double limit)
{
    RegBounds out = in;
    for (int r = 0; r < 16; r++)
        if (in.v[r] > limit) out.v[r] = kReduced;
    return out;
}
constexpr RegBounds reduce_mask_bounds(const RegBounds& in, unsigned mask)
{
    RegBounds out = in;
    for (int r = 0; r < 16; r++)
        if ((mask >> r) & 1u) out.v[r] = kReduced;
    return out;
}

// ---- schedules as types: S::in() is the bound of every register on entry ----
template <class S, bool HI>
struct AfterCt {
    static constexpr RegBounds in() { return ct_bounds<HI>(S::in()); }
};
template <class S, bool HI>
struct AfterGs {
    static constexpr RegBounds in() { return gs_bounds<HI>(S::in()); }
};
template <class S>
struct AfterXposeAll {
    static constexpr RegBounds in() { return xpose_all_bounds(S::in()); }
};
template <class S>
struct AfterXposeBC {
    static constexpr RegBounds in() { return xpose_bc_bounds(S::in()); }
};
template <class S>
struct AfterXposeCB {
    static constexpr RegBounds in() { return xpose_cb_bounds(S::in()); }
};
// LIMIT_MILLI: limit in thousandths of p (a double cannot be a template argument before C++20)
template <class S, int LIMIT_MILLI>
struct AfterReduce {
    static constexpr RegBounds in() { return reduce_bounds(S::in(), LIMIT_MILLI * 0.001); }
};
template <class S, unsigned MASK>
struct AfterReduceMask {
    static constexpr RegBounds in() { return reduce_mask_bounds(S::in(), MASK); }
};
template <int MILLI>
struct Uniform {
    static constexpr RegBounds in() { return uniform(MILLI * 0.001); }
};

// ---- the schedules of the 1024-point transforms of ntt_wave.h (layouts A, B, C there) ----
// bounds after the exact first two stages of a polynomial of integers |x| <= DIGIT_MAX
template <int DIGIT_MAX>
struct DigitsAfterExact {
    static constexpr RegBounds in()
    {
        const double s1 = DIGIT_MAX * (1.0 + fpf::ROOT4 + fpf::ROOT8 + fpf::ROOT8 * fpf::ROOT8 * fpf::ROOT8);
        return uniform(s1 < 9007199254740992.0 / 64.0 ? s1 / fpf::P : kPoison);      // six bits of headroom below 2^53, as forward_digit_spectrum_bound
    }
};
// Forward transform of a gadget-digit polynomial, phase A (stages 0-3, natural order in), and the schedule it leaves
template <int DIGIT_MAX>
struct FwdDigits {
    using A0 = DigitsAfterExact<DIGIT_MAX>;
    using B0 = AfterXposeAll<AfterCt<A0, false>>;                    // after stages 2-3 and A -> B
    using B1 = AfterCt<B0, true>;                                         // after stages 4-5
    using C0 = AfterXposeBC<AfterCt<B1, false>>;                      // after stages 6-7 and B -> C
    // Stages 8-9: the pass-through register of each group has collected 3 x 3 reduced addends by now and takes 3 more
    // here: it is reduced first (4 registers, 12 operations; the rest of the polynomial never is)
    static constexpr unsigned kReduceC = 0x1111u;
    using C1 = AfterReduceMask<C0, kReduceC>;
    using Spectrum = AfterCt<C1, false>;
};
// `ROWS` products of a spectrum with schedule SPEC summed into one accumulator register by register; FUSED: the last
// one is mulmod_add (it reduces the whole sum, fpfield.h), otherwise the caller reduces
template <class SPEC, int ROWS, bool FUSED>
struct PointwiseSum {
    static constexpr RegBounds in()
    {
        RegBounds out{};
        const RegBounds s = SPEC::in();
        for (int r = 0; r < 16; r++) {
            const double prod = after_product(s.v[r]);
            if (FUSED) {
                const double last = s.v[r] >= fpf::LIM_WIDE ? kPoison : needs_wide(s.v[r]) ? fpf::after_mulmod_add_wide(s.v[r]) : fpf::after_mulmod_add(s.v[r]);
                out.v[r] = checked((ROWS - 1) * prod + last) >= kPoison ? kPoison : last;     // |h - q p| <= |c| + |result|
            } else {
                out.v[r] = checked(ROWS * prod);
            }
        }
        return out;
    }
};

// Inverse transform: x in layout C with the bounds of schedule S0 (reduced sums: Uniform<501>; sums whose last
// product was mulmod_add: PointwiseSum<.., true>), out in layout A, NOT scaled by 1/N.  Out::in() are the output bounds
// per register (lift_add below picks the lift by them).
template <class S0>
struct Inverse {
    static constexpr int kLimit = 2570;                                      // registers above 2^51 are reduced between passes
    using C1 = AfterGs<S0, false>;                                        // after stages 9-8
    using B0 = AfterXposeCB<AfterReduce<C1, kLimit>>;
    using B1 = AfterReduce<AfterGs<B0, false>, kLimit>;               // after stages 7-6
    using B2 = AfterReduce<AfterGs<B1, true>, kLimit>;                // after stages 5-4
    using A0 = AfterXposeAll<B2>;
    using A1 = AfterReduce<AfterGs<A0, false>, kLimit>;               // after stages 3-2
    using Out = AfterGs<A1, true>;                                        // after stages 1-0
};

// ---- the passes -------------------------------------------------------------------------------------------------
template <bool WIDE>
R4_HD double product(double a, double w) { return WIDE ? fpf::mulmod_wide(a, w) : fpf::mulmod(a, w); }

template <bool WB, bool WC, bool WD>
R4_HD void ct_bfly4(double& x0, double& xf, double& xc, double& xcf, double w, double u, double uw)
{
    const double b = product<WB>(xc, w), c = product<WC>(xf, u), d = product<WD>(xcf, uw);
    const double s = x0 + b, e = x0 - b, t = c + d, v = fpf::mul_root4(c - d);
    x0 = s + t;
    xf = s - t;
    xc = e + v;
    xcf = e - v;
}
template <bool WC, bool WF, bool WCF>
R4_HD void gs_bfly4(double& x0, double& xf, double& xc, double& xcf, double w, double v, double vw)
{
    const double s0 = x0 + xf, p = x0 - xf, s1 = xc + xcf, q = fpf::mul_root4(xc - xcf);
    x0 = s0 + s1;
    xc = product<WC>(s0 - s1, w);
    xf = product<WF>(p - q, v);
    xcf = product<WCF>(p + q, vw);
}

// One group of a pass, the wide / narrow choice of each product read from the schedule at compile time.
template <class S, bool HI, int G>
struct Group {
    static constexpr RegBounds b = S::in();
    static constexpr int r0 = reg_of<HI>(G, 0, 0), rf = reg_of<HI>(G, 0, 1), rc = reg_of<HI>(G, 1, 0), rcf = reg_of<HI>(G, 1, 1);
    static R4_HD void ct(double (&x)[16], double w, double u, double uw)
    {
        ct_bfly4<needs_wide(b.v[rc]), needs_wide(b.v[rf]), needs_wide(b.v[rcf])>(x[r0], x[rf], x[rc], x[rcf], w, u, uw);
    }
    static R4_HD void gs(double (&x)[16], double w, double v, double vw)
    {
        constexpr double s0 = b.v[r0] + b.v[rf], s1 = b.v[rc] + b.v[rcf];
        gs_bfly4<needs_wide(s0 + s1), needs_wide(s0 + fpf::AFTER_MUL_ROOT4), needs_wide(s0 + fpf::AFTER_MUL_ROOT4)>(x[r0], x[rf], x[rc], x[rcf], w, v, vw);
    }
};

// Twiddles of a 16-register block tw(0..14) = [w | u, u w | w_0..w_3 | u_0, u_0 w_0, .. u_3, u_3 w_3] (r4 tables).
// HI pass: one twiddle triple for all four groups; LO pass: group g takes tw(WB + g), tw(UB + 2g), tw(UB + 2g + 1).
template <class S, class TW>
R4_HD void ct_pass_hi(double (&x)[16], const TW& tw)
{
    const double w = tw(0), u = tw(1), uw = tw(2);
    Group<S, true, 0>::ct(x, w, u, uw);
    Group<S, true, 1>::ct(x, w, u, uw);
    Group<S, true, 2>::ct(x, w, u, uw);
    Group<S, true, 3>::ct(x, w, u, uw);
}
template <class S, int WB, int UB, class TW>
R4_HD void ct_pass_lo(double (&x)[16], const TW& tw)
{
    Group<S, false, 0>::ct(x, tw(WB + 0), tw(UB + 0), tw(UB + 1));
    Group<S, false, 1>::ct(x, tw(WB + 1), tw(UB + 2), tw(UB + 3));
    Group<S, false, 2>::ct(x, tw(WB + 2), tw(UB + 4), tw(UB + 5));
    Group<S, false, 3>::ct(x, tw(WB + 3), tw(UB + 6), tw(UB + 7));
}
template <class S, class TW>
R4_HD void gs_pass_hi(double (&x)[16], const TW& tw)
{
    const double w = tw(0), v = tw(1), vw = tw(2);
    Group<S, true, 0>::gs(x, w, v, vw);
    Group<S, true, 1>::gs(x, w, v, vw);
    Group<S, true, 2>::gs(x, w, v, vw);
    Group<S, true, 3>::gs(x, w, v, vw);
}
template <class S, int WB, int UB, class TW>
R4_HD void gs_pass_lo(double (&x)[16], const TW& tw)
{
    Group<S, false, 0>::gs(x, tw(WB + 0), tw(UB + 0), tw(UB + 1));
    Group<S, false, 1>::gs(x, tw(WB + 1), tw(UB + 2), tw(UB + 3));
    Group<S, false, 2>::gs(x, tw(WB + 2), tw(UB + 4), tw(UB + 5));
    Group<S, false, 3>::gs(x, tw(WB + 3), tw(UB + 6), tw(UB + 7));
}
// reduce the registers whose bound exceeds the limit (the ones AfterReduce<S, LIMIT_MILLI> resets)
template <class S, int LIMIT_MILLI, int R = 0>
R4_HD void reduce_above(double (&x)[16])
{
    if constexpr (R < 16) {
        if constexpr (S::in().v[R] > LIMIT_MILLI * 0.001) x[R] = fpf::reduce(x[R]);
        reduce_above<S, LIMIT_MILLI, R + 1>(x);
    }
}

// the same for the four registers of HI group G (registers G, G + 4, G + 8, G + 12)
template <class S, int LIMIT_MILLI, int G>
R4_HD void reduce_above_group(double (&x)[16])
{
    if constexpr (S::in().v[G] > LIMIT_MILLI * 0.001) x[G] = fpf::reduce(x[G]);
    if constexpr (S::in().v[G + 4] > LIMIT_MILLI * 0.001) x[G + 4] = fpf::reduce(x[G + 4]);
    if constexpr (S::in().v[G + 8] > LIMIT_MILLI * 0.001) x[G + 8] = fpf::reduce(x[G + 8]);
    if constexpr (S::in().v[G + 12] > LIMIT_MILLI * 0.001) x[G + 12] = fpf::reduce(x[G + 12]);
}
template <unsigned MASK, int R = 0>
R4_HD void reduce_mask(double (&x)[16])
{
    if constexpr (R < 16) {
        if constexpr (((MASK >> R) & 1u) != 0) x[R] = fpf::reduce(x[R]);
        reduce_mask<MASK, R + 1>(x);
    }
}

}  // namespace r4
}  // namespace cufhe_amd
