// host_model.cpp -- host-side model of the device arithmetic (test infrastructure).
//
// Compiles cufhe_amd/csrc/fpfield.h with g++ and replays, on scalar loops, exactly the
// lazy-reduction schedule of cufhe_amd/csrc/ntt_wave.h (which stages use mulmod,
// mulmod_wide, reduce).  It checks what cannot be seen from GPU parity tests alone:
//   - every intermediate stays an exact integer below 2^53 and every mulmod input
//     stays inside its documented range, for random AND adversarial inputs;
//   - the per-stage magnitude bounds written in ntt_wave.h hold;
//   - the result equals the exact negacyclic product (checked by the caller against the
//     oracle's schoolbook product).
// Index order here is the plain radix-2 order (no lane layouts): the layouts of the device
// code only permute which register holds which element.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../cufhe_amd/csrc/fpfield.h"
#include "../../cufhe_amd/csrc/ntt_r4.h"

namespace {
typedef unsigned __int128 u128;
constexpr int N = 1024;
const uint64_t P = fpf::P_U64;

uint64_t mulmod_u(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % P); }
uint64_t powmod_u(uint64_t a, uint64_t e)
{
    uint64_t r = 1;
    while (e) { if (e & 1) r = mulmod_u(r, a); a = mulmod_u(a, a); e >>= 1; }
    return r;
}
double bal(uint64_t v) { return v > P / 2 ? -(double)(P - v) : (double)v; }
uint32_t bitrev10(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 10; i++) r |= ((x >> i) & 1u) << (9 - i); return r; }

std::vector<double> g_fwd, g_inv;
double g_ninv;
void tables()
{
    if (!g_fwd.empty()) return;
    g_fwd.resize(N); g_inv.resize(N);
    uint64_t psi = fpf::PSI_2048, psi_inv = powmod_u(psi, P - 2);
    for (uint32_t i = 0; i < N; i++) { g_fwd[i] = bal(powmod_u(psi, bitrev10(i))); g_inv[i] = bal(powmod_u(psi_inv, bitrev10(i))); }
    g_ninv = bal(powmod_u(N, P - 2));
}

struct Track {           // violation flags + per-stage maxima (units of p)
    double max_abs = 0;  // largest |value| seen anywhere / p
    double max_mul_in = 0, max_wide_in = 0;
    int bad = 0;         // non-integer / out-of-range events
    void val(double v)
    {
        double a = std::fabs(v);
        if (a >= 9007199254740992.0 || v != std::nearbyint(v)) bad++;
        if (a / fpf::P > max_abs) max_abs = a / fpf::P;
    }
    void mul_in(double v, bool wide)
    {
        double a = std::fabs(v);
        if (wide) { if (a >= 9007199254740992.0) bad++; if (a / fpf::P > max_wide_in) max_wide_in = a / fpf::P; }
        else { if (a >= 4503599627370496.0) bad++; if (a / fpf::P > max_mul_in) max_mul_in = a / fpf::P; }
    }
};

double mm(double a, double w, bool wide, Track& t)
{
    t.mul_in(a, wide);
    double r = wide ? fpf::mulmod_wide(a, w) : fpf::mulmod(a, w);
    t.val(r);
    return r;
}

// forward (ntt_wave.h ntt_forward<SMALL_IN>): small_in (gadget digits): stages 0 and 1 are ONE exact radix-4 butterfly on the
// original inputs (roots I, zeta and zeta^3 I = -zeta: nothing is reduced), stages 2..8 mulmod, stage 9
// wide; otherwise (32-bit words) stages 0..7 mulmod, 8 and 9 wide.
void fwd(double* x, Track& t, double* stage_max, bool small_in)
{
    int tt = N >> 1, s = 0, m = 1;
    if (small_in) {
        // ct_four_stages<SMALL_IN>: elements e, e + 256, e + 512, e + 768
        const double Z3 = fpf::ROOT8 * fpf::ROOT8 * fpf::ROOT8;
        if (g_fwd[1] != fpf::ROOT4 || g_fwd[2] != fpf::ROOT8 || g_fwd[3] != Z3) t.bad++;      // the table's roots are the small ones the kernel hard-codes
        double mx = 0;
        for (int j = 0; j < N / 4; j++) {
            const double a = x[j], a1 = x[j + N / 4], b = x[j + N / 2], b1 = x[j + 3 * N / 4];
            const double u = std::fma(b, fpf::ROOT4, a), v = std::fma(-b, fpf::ROOT4, a);
            const double u1 = std::fma(b1, fpf::ROOT4, a1);
            const double p0 = b1 * fpf::ROOT8;
            const double tz = std::fma(a1, Z3, p0);
            const double q0 = std::fma(u1, fpf::ROOT8, u), q1 = std::fma(-u1, fpf::ROOT8, u), q2 = v + tz, q3 = v - tz;
            for (double w : {u, v, u1, p0, tz, q0, q1, q2, q3}) t.val(w);
            // the same values from the two plain stages in exact integer arithmetic (everything is far below 2^63)
            const long long A = (long long)a, A1 = (long long)a1, B = (long long)b, B1 = (long long)b1, I = (long long)fpf::ROOT4, Z = (long long)fpf::ROOT8;
            const long long U = A + I * B, V = A - I * B, U1 = A1 + I * B1, V1 = A1 - I * B1;
            const long long Z3m = (long long)Z3;
            // zeta^3 V1 = zeta^3 A1 - zeta^3 I B1 and zeta^3 I = zeta^5 = -zeta (mod p): compare modulo p
            auto modp = [](__int128 z) { long long r = (long long)(z % (__int128)fpf::P_U64); return r < 0 ? r + (long long)fpf::P_U64 : r; };
            if (modp((__int128)Z3m * V1) != modp((__int128)Z3m * A1 + (__int128)Z * B1)) t.bad++;
            if ((long long)q0 != U + Z * U1 || (long long)q1 != U - Z * U1) t.bad++;
            if (modp((__int128)(long long)q2) != modp((__int128)V + (__int128)Z3m * V1) || modp((__int128)(long long)q3) != modp((__int128)V - (__int128)Z3m * V1)) t.bad++;
            x[j] = q0; x[j + N / 4] = q1; x[j + N / 2] = q2; x[j + 3 * N / 4] = q3;
            mx = std::fmax(mx, std::fmax(std::fmax(std::fabs(q0), std::fabs(q1)), std::fmax(std::fabs(q2), std::fabs(q3))));
        }
        if (stage_max) stage_max[0] = stage_max[1] = mx / fpf::P;
        m = 4; tt = N >> 3; s = 2;
    }
    for (; m < N; m <<= 1, tt >>= 1, s++) {
        const bool wide = small_in ? s >= 9 : s >= 8;
        double mx = 0;
        for (int g = 0; g < m; g++) {
            const double w = g_fwd[m + g];
            double* a = x + 2 * g * tt;
            for (int j = 0; j < tt; j++) {
                double u = a[j];
                const double v = mm(a[j + tt], w, wide, t);
                a[j] = u + v; a[j + tt] = u - v;
                t.val(a[j]); t.val(a[j + tt]);
                mx = std::fmax(mx, std::fmax(std::fabs(a[j]), std::fabs(a[j + tt])));
            }
        }
        if (stage_max) stage_max[s] = mx / fpf::P;
    }
}
// inverse: s9 s8 | s7 s6(wide) reduce s5 s4 | s3 s2(wide) reduce s1 s0   (ntt_wave.h ntt_inverse)
void inv(double* x, Track& t, double* stage_max)
{
    int tt = 1, s = 9;
    for (int m = N >> 1; m >= 1; m >>= 1, tt <<= 1, s--) {
        const bool wide = (s == 6 || s == 2);
        double mx = 0;
        for (int g = 0; g < m; g++) {
            const double w = g_inv[m + g];
            double* a = x + 2 * g * tt;
            for (int j = 0; j < tt; j++) {
                double u = a[j], v = a[j + tt];
                a[j] = u + v; t.val(a[j]);
                double d = u - v; t.val(d);
                a[j + tt] = mm(d, w, wide, t);
                mx = std::fmax(mx, std::fmax(std::fabs(a[j]), std::fabs(a[j + tt])));
            }
        }
        if (stage_max) stage_max[9 - s] = mx / fpf::P;
        if (s == 6 || s == 2)
            for (int i = 0; i < N; i++) { x[i] = fpf::reduce(x[i]); t.val(x[i]); }
    }
}
}  // namespace


// ---- radix-4 schedule (cufhe_amd/csrc/ntt_r4.h, ntt_wave.h: ntt_forward_digits_*_r4 / ntt_inverse_r4) ----------------
// The pass functions are the device's own (ntt_r4.h is host-callable); what is emulated here are the 64 lanes and the three
// register layouts of ntt_wave.h, element index e = position in the in-place array:
//   A: lane = e[5:0], reg = e[9:6]     B: lane = e[9:6] | e[1:0] << 4, reg = e[5:2]     C: lane = e[9:6] | e[5:4] << 4, reg = e[3:0]
// After every pass each value must be an integer below 2^53 AND below the compile-time bound of its register (Sched::in()).
namespace r4m {
using namespace cufhe_amd::r4;
int elem(char layout, int lane, int reg)
{
    const int lam = lane & 15, hi = lane >> 4;
    if (layout == 'A') return lane | reg << 6;
    if (layout == 'B') return lam << 6 | reg << 2 | hi;
    return lam << 6 | hi << 4 | reg;
}
// r4 tables, written down from the index formulas of capi.hip (fill_tables) independently of that code
double mulb(double a, double b)
{
    const uint64_t ua = a < 0 ? P - (uint64_t)(-a) : (uint64_t)a, ub = b < 0 ? P - (uint64_t)(-b) : (uint64_t)b;
    return bal(mulmod_u(ua, ub));
}
struct Block { double t[15]; double operator()(int k) const { return t[k]; } };
Block block_r4(const std::vector<double>& root, int base, int lam)     // radix-2 block: root[base * 2^lvl + lam * 2^lvl + j]
{
    Block b;
    for (int k = 0; k < 15; k++) {
        int lvl = 0;
        while ((2 << lvl) <= k + 1) lvl++;
        b.t[k] = root[(base << lvl) + (lam << lvl) + (k + 1 - (1 << lvl))];
    }
    b.t[2] = mulb(b.t[1], b.t[0]);
    for (int g = 0; g < 4; g++) b.t[8 + 2 * g] = mulb(b.t[7 + 2 * g], b.t[3 + g]);
    return b;
}
struct BlockC { double t[12]; double operator()(int k) const { return t[k]; } };
BlockC block_c_r4(const std::vector<double>& root, int lane)
{
    BlockC b;
    const int lam = lane & 15, h = lane >> 4;
    for (int k = 0; k < 12; k++) b.t[k] = k < 4 ? root[256 + ((lam << 4) | (h << 2) | k)] : root[512 + ((lam << 5) | (h << 3) | (k - 4))];
    for (int g = 0; g < 4; g++) b.t[5 + 2 * g] = mulb(b.t[4 + 2 * g], b.t[g]);
    return b;
}
struct Check {
    int bad = 0;
    double worst = 0;       // largest value / (bound of its register)
    template <class S>
    void regs(const double (&x)[16])
    {
        constexpr RegBounds b = S::in();
        for (int r = 0; r < 16; r++) {
            const double a = std::fabs(x[r]);
            if (a >= 9007199254740992.0 || x[r] != std::nearbyint(x[r])) bad++;
            if (a > b.v[r] * fpf::P) bad++;
            if (a / (b.v[r] * fpf::P) > worst) worst = a / (b.v[r] * fpf::P);
        }
    }
};
template <class F>
void per_lane(double* X, char layout, F f)
{
    for (int lane = 0; lane < 64; lane++) {
        double x[16];
        for (int r = 0; r < 16; r++) x[r] = X[elem(layout, lane, r)];
        f(x, lane);
        for (int r = 0; r < 16; r++) X[elem(layout, lane, r)] = x[r];
    }
}
template <int DIGIT_MAX>
void forward_digits(double* X, Check& ck)
{
    using F = FwdDigits<DIGIT_MAX>;
    static_assert(valid(F::Spectrum::in()), "forward schedule");
    const double Z3 = fpf::ROOT8 * fpf::ROOT8 * fpf::ROOT8;
    per_lane(X, 'A', [&](double (&x)[16], int) {
        for (int r = 0; r < 4; r++) {            // ct_exact_first_two
            const double a = x[r], a1 = x[r + 4], b = x[r + 8], b1 = x[r + 12];
            const double u = std::fma(b, fpf::ROOT4, a), v = std::fma(-b, fpf::ROOT4, a);
            const double u1 = std::fma(b1, fpf::ROOT4, a1);
            const double t = std::fma(a1, Z3, b1 * fpf::ROOT8);
            x[r] = std::fma(u1, fpf::ROOT8, u); x[r + 4] = std::fma(-u1, fpf::ROOT8, u); x[r + 8] = v + t; x[r + 12] = v - t;
        }
        ck.regs<typename F::A0>(x);
        ct_pass_lo<typename F::A0, 3, 7>(x, block_r4(g_fwd, 1, 0));
        ck.regs<AfterCt<typename F::A0, false>>(x);
    });
    per_lane(X, 'B', [&](double (&x)[16], int lane) {
        const Block tw = block_r4(g_fwd, 16, lane & 15);
        ck.regs<typename F::B0>(x);
        ct_pass_hi<typename F::B0>(x, tw);
        ck.regs<typename F::B1>(x);
        ct_pass_lo<typename F::B1, 3, 7>(x, tw);
        ck.regs<AfterCt<typename F::B1, false>>(x);
    });
    per_lane(X, 'C', [&](double (&x)[16], int lane) {
        ck.regs<typename F::C0>(x);
        reduce_mask<F::kReduceC>(x);
        ck.regs<typename F::C1>(x);
        ct_pass_lo<typename F::C1, 0, 4>(x, block_c_r4(g_fwd, lane));
        ck.regs<typename F::Spectrum>(x);
    });
}
template <class S0>
void inverse(double* X, Check& ck)
{
    using V = Inverse<S0>;
    static_assert(valid(V::Out::in()), "inverse schedule");
    per_lane(X, 'C', [&](double (&x)[16], int lane) {
        ck.regs<S0>(x);
        gs_pass_lo<S0, 0, 4>(x, block_c_r4(g_inv, lane));
        ck.regs<typename V::C1>(x);
        reduce_above<typename V::C1, V::kLimit>(x);
    });
    per_lane(X, 'B', [&](double (&x)[16], int lane) {
        const Block tw = block_r4(g_inv, 16, lane & 15);
        ck.regs<typename V::B0>(x);
        gs_pass_lo<typename V::B0, 3, 7>(x, tw);
        ck.regs<AfterGs<typename V::B0, false>>(x);
        reduce_above<AfterGs<typename V::B0, false>, V::kLimit>(x);
        ck.regs<typename V::B1>(x);
        gs_pass_hi<typename V::B1>(x, tw);
        ck.regs<AfterGs<typename V::B1, true>>(x);
        reduce_above<AfterGs<typename V::B1, true>, V::kLimit>(x);
        ck.regs<typename V::B2>(x);
    });
    per_lane(X, 'A', [&](double (&x)[16], int) {
        const Block tw = block_r4(g_inv, 1, 0);
        ck.regs<typename V::A0>(x);
        gs_pass_lo<typename V::A0, 3, 7>(x, tw);
        ck.regs<AfterGs<typename V::A0, false>>(x);
        reduce_above<AfterGs<typename V::A0, false>, V::kLimit>(x);
        ck.regs<typename V::A1>(x);
        gs_pass_hi<typename V::A1>(x, tw);
        ck.regs<typename V::Out>(x);
    });
}
// the lift blind_rotate_kernel's lift_add picks per register (layout A: reg = e >> 6)
template <class OUT>
uint32_t lift(double v, int reg, Check& ck)
{
    constexpr RegBounds b = OUT::in();
    if (b.v[reg] < 2.57) {
        if (std::fabs(v) >= 2251799813685248.0) ck.bad++;
        return fpf::lift_u32_small(v);
    }
    return fpf::lift_u32(v);
}
}  // namespace r4m

extern "C" {

// scalar checks: returns number of mismatches against exact arithmetic
int hm_check_mulmod(const double* a, const double* w, int count, int wide)
{
    int bad = 0;
    for (int i = 0; i < count; i++) {
        double r = wide ? fpf::mulmod_wide(a[i], w[i]) : fpf::mulmod(a[i], w[i]);
        // exact: (a*w - r) must be divisible by p and |r| within the documented bound
        __int128 prod = (__int128)(int64_t)a[i] * (int64_t)w[i];
        __int128 diff = prod - (__int128)(int64_t)r;
        if (diff % (__int128)P != 0) bad++;
        double c = std::fabs(a[i]) / fpf::P;
        double bound = (wide ? 1.0 : 0.5) + 0.0973 * c + 1e-9;
        if (std::fabs(r) > bound * fpf::P) bad++;
        if (r != std::nearbyint(r)) bad++;
    }
    return bad;
}
int hm_check_reduce_lift(const double* a, int count)
{
    int bad = 0;
    for (int i = 0; i < count; i++) {
        double r = fpf::reduce(a[i]);
        __int128 diff = (__int128)(int64_t)a[i] - (__int128)(int64_t)r;
        if (diff % (__int128)P != 0) bad++;
        if (std::fabs(r) > 0.5 * fpf::P + 1) bad++;
        int64_t c = (int64_t)r;
        if (fpf::low32(r) != (uint32_t)(uint64_t)c) bad++;
        if (fpf::lift_u32(a[i]) != (uint32_t)(uint64_t)c) bad++;
        if (std::fabs(a[i]) < 2251799813685248.0 && fpf::lift_u32_small(a[i]) != (uint32_t)(uint64_t)c) bad++;
    }
    return bad;
}

// res = a * b negacyclic mod 2^32 through the device schedule.
// stats[0]=bad events, [1]=max|v|/p, [2]=max mulmod input/p, [3]=max wide input/p,
// [4..13] forward per-stage maxima of the digit transform, [14..23] inverse per-stage maxima.
void hm_polymul(uint32_t* res, const int32_t* a, const uint32_t* b, double* stats)
{
    tables();
    Track t;
    std::vector<double> x(N), y(N);
    for (int i = 0; i < N; i++) { x[i] = (double)a[i]; y[i] = (double)(int32_t)b[i]; }
    bool small = true;
    for (int i = 0; i < N; i++) small = small && a[i] >= -512 && a[i] <= 512;      // gadget digits up to Bg = 2^10 take the exact first stages (kernels_ps.hip.h: small_ok)
    fwd(x.data(), t, stats ? stats + 4 : nullptr, small);
    fwd(y.data(), t, nullptr, false);
    for (int i = 0; i < N; i++) {
        // BK conversion: scale by N^-1 and centre (bk_to_ntt_kernel)
        y[i] = fpf::reduce(mm(y[i], g_ninv, true, t));
        // pointwise (pointwise_accumulate) + pre-inverse reduce (inverse_and_add)
        x[i] = fpf::reduce(mm(x[i], y[i], true, t));
        t.val(x[i]);
    }
    inv(x.data(), t, stats ? stats + 14 : nullptr);
    for (int i = 0; i < N; i++) res[i] = fpf::lift_u32(x[i]);
    if (stats) { stats[0] = t.bad; stats[1] = t.max_abs; stats[2] = t.max_mul_in; stats[3] = t.max_wide_in; }
}

// One external product with the device's accumulation schedule: rows = 6 digit polys
// dig[6][N] (signed), bk[6][2][N] torus words -> out[2][N] torus words (the value added to
// the accumulator).  Mirrors cmux_component / inverse_and_add (six products, one reduce).
void hm_external_product(uint32_t* out, const int32_t* dig, const uint32_t* bk, double* stats)
{
    tables();
    Track t;
    std::vector<double> A0(N, 0.0), A1(N, 0.0), x(N), y0(N), y1(N);
    for (int row = 0; row < 6; row++) {
        for (int i = 0; i < N; i++) {
            x[i] = (double)dig[row * N + i];
            y0[i] = (double)(int32_t)bk[(row * 2 + 0) * N + i];
            y1[i] = (double)(int32_t)bk[(row * 2 + 1) * N + i];
        }
        fwd(x.data(), t, nullptr, true); fwd(y0.data(), t, nullptr, false); fwd(y1.data(), t, nullptr, false);
        for (int i = 0; i < N; i++) {
            y0[i] = fpf::reduce(mm(y0[i], g_ninv, true, t));
            y1[i] = fpf::reduce(mm(y1[i], g_ninv, true, t));
            A0[i] += mm(x[i], y0[i], true, t); t.val(A0[i]);
            A1[i] += mm(x[i], y1[i], true, t); t.val(A1[i]);
        }
    }
    for (int i = 0; i < N; i++) { A0[i] = fpf::reduce(A0[i]); A1[i] = fpf::reduce(A1[i]); }
    inv(A0.data(), t, nullptr); inv(A1.data(), t, nullptr);
    for (int i = 0; i < N; i++) { out[i] = fpf::lift_u32(A0[i]); out[N + i] = fpf::lift_u32(A1[i]); }
    if (stats) { stats[0] = t.bad; stats[1] = t.max_abs; stats[2] = t.max_mul_in; stats[3] = t.max_wide_in; }
}

// The same external product with the schedule of the low-latency kernel
// (cufhe_amd/csrc/ntt_wave512.h, kernels_ll.hip.h): exact first stage u_h = a[e] +- I a[e+512],
// two 512-point half transforms (stages 1..8 narrow, stage 9 wide), six wide products per sum,
// reduce, half inverses (s9 s8 s7 | s6 wide, reduce, s5 s4 | s3, s2 wide, reduce, s1), last stage
// (u0 + u1, (u0 - u1)(-I)), lift_u32_small.  stats as above.
void hm_external_product_split(uint32_t* out, const int32_t* dig, const uint32_t* bk, double* stats)
{
    tables();
    Track t;
    constexpr int H = N / 2;
    std::vector<double> A0(N, 0.0), A1(N, 0.0), x(N), y0(N), y1(N);
    for (int row = 0; row < 6; row++) {
        for (int i = 0; i < N; i++) {
            y0[i] = (double)(int32_t)bk[(row * 2 + 0) * N + i];
            y1[i] = (double)(int32_t)bk[(row * 2 + 1) * N + i];
        }
        fwd(y0.data(), t, nullptr, false); fwd(y1.data(), t, nullptr, false);   // key conversion: full transform
        if (g_fwd[1] != fpf::ROOT4) t.bad++;
        for (int e = 0; e < H; e++) {                                           // first stage, exact
            const double a = (double)dig[row * N + e], b = (double)dig[row * N + e + H];
            x[e] = __builtin_fma(b, fpf::ROOT4, a);
            x[e + H] = __builtin_fma(-b, fpf::ROOT4, a);
            t.val(x[e]); t.val(x[e + H]);
        }
        for (int h = 0; h < 2; h++) {                                           // stages 1..9 of half h
            double* xh = x.data() + h * H;
            int tt = H >> 1, s = 1;
            for (int m = 1; m < H; m <<= 1, tt >>= 1, s++)
                for (int g = 0; g < m; g++) {
                    const double w = g_fwd[2 * m + h * m + g];
                    double* a = xh + 2 * g * tt;
                    for (int j = 0; j < tt; j++) {
                        const double u = a[j], v = mm(a[j + tt], w, s == 9, t);
                        a[j] = u + v; a[j + tt] = u - v;
                        t.val(a[j]); t.val(a[j + tt]);
                    }
                }
        }
        for (int i = 0; i < N; i++) {
            y0[i] = fpf::reduce(mm(y0[i], g_ninv, true, t));
            y1[i] = fpf::reduce(mm(y1[i], g_ninv, true, t));
            A0[i] += mm(x[i], y0[i], true, t); t.val(A0[i]);
            A1[i] += mm(x[i], y1[i], true, t); t.val(A1[i]);
        }
    }
    for (int o = 0; o < 2; o++) {
        std::vector<double>& A = o ? A1 : A0;
        for (int i = 0; i < N; i++) A[i] = fpf::reduce(A[i]);
        for (int h = 0; h < 2; h++) {
            double* xh = A.data() + h * H;
            int tt = 1, s = 9;
            for (int m = H >> 1; m >= 1; m >>= 1, tt <<= 1, s--) {
                const bool wide = (s == 6 || s == 2);
                for (int g = 0; g < m; g++) {
                    const double w = g_inv[2 * m + h * m + g];
                    double* a = xh + 2 * g * tt;
                    for (int j = 0; j < tt; j++) {
                        const double u = a[j], v = a[j + tt];
                        a[j] = u + v; t.val(a[j]);
                        const double d = u - v; t.val(d);
                        a[j + tt] = mm(d, w, wide, t);
                    }
                }
                if (wide) for (int i = 0; i < H; i++) { xh[i] = fpf::reduce(xh[i]); t.val(xh[i]); }
            }
        }
        if (g_inv[1] != -fpf::ROOT4) t.bad++;
        for (int e = 0; e < H; e++) {
            const double u0 = A[e], u1 = A[e + H];
            const double lo = u0 + u1; t.val(lo);
            const double d = u0 - u1; t.val(d);
            const double hi = mm(d, -fpf::ROOT4, false, t);
            if (std::fabs(lo) >= 2251799813685248.0 || std::fabs(hi) >= 2251799813685248.0) t.bad++;   // lift_u32_small domain
            out[o * N + e] = fpf::lift_u32_small(lo);
            out[o * N + e + H] = fpf::lift_u32_small(hi);
        }
    }
    if (stats) { stats[0] = t.bad; stats[1] = t.max_abs; stats[2] = t.max_mul_in; stats[3] = t.max_wide_in; }
}

// One external product with the schedule of blind_rotate_kernel (radix-4 passes, per-register bounds, the last
// product of each sum reducing it): same interface as hm_external_product; stats[4] = largest value / its register's
// compile-time bound (must stay <= 1), stats[5..20] = the spectrum bound per layout-C register, stats[21..36] = the inverse's
// output bound per layout-A register.
void hm_external_product_r4(uint32_t* out, const int32_t* dig, const uint32_t* bk, double* stats)
{
    using namespace r4m;
    tables();
    Track t;
    Check ck;
    constexpr int kDigitMax = 32;
    using Spec = FwdDigits<kDigitMax>::Spectrum;
    using Sums = PointwiseSum<Spec, 6, true>;
    using Out = Inverse<Sums>::Out;
    constexpr RegBounds sb = Spec::in();
    std::vector<double> A0(N, 0.0), A1(N, 0.0), x(N), y0(N), y1(N);
    for (int row = 0; row < 6; row++) {
        for (int i = 0; i < N; i++) {
            x[i] = (double)dig[row * N + i];
            if (std::abs(dig[row * N + i]) > kDigitMax) ck.bad++;
            y0[i] = (double)(int32_t)bk[(row * 2 + 0) * N + i];
            y1[i] = (double)(int32_t)bk[(row * 2 + 1) * N + i];
        }
        forward_digits<kDigitMax>(x.data(), ck);
        fwd(y0.data(), t, nullptr, false); fwd(y1.data(), t, nullptr, false);      // the key is still converted by the radix-2 transform
        for (int i = 0; i < N; i++) {
            y0[i] = fpf::reduce(mm(y0[i], g_ninv, true, t));
            y1[i] = fpf::reduce(mm(y1[i], g_ninv, true, t));
            const bool wide = needs_wide(sb.v[i & 15]);                             // layout C: reg = e[3:0]
            if (row == 5) {
                A0[i] = wide ? fpf::mulmod_add_wide(x[i], y0[i], A0[i]) : fpf::mulmod_add(x[i], y0[i], A0[i]);
                A1[i] = wide ? fpf::mulmod_add_wide(x[i], y1[i], A1[i]) : fpf::mulmod_add(x[i], y1[i], A1[i]);
            } else {
                A0[i] += wide ? fpf::mulmod_wide(x[i], y0[i]) : fpf::mulmod(x[i], y0[i]);
                A1[i] += wide ? fpf::mulmod_wide(x[i], y1[i]) : fpf::mulmod(x[i], y1[i]);
            }
            t.val(A0[i]); t.val(A1[i]);
        }
    }
    inverse<Sums>(A0.data(), ck); inverse<Sums>(A1.data(), ck);
    for (int i = 0; i < N; i++) { out[i] = lift<Out>(A0[i], i >> 6, ck); out[N + i] = lift<Out>(A1[i], i >> 6, ck); }
    if (stats) {
        stats[0] = t.bad + ck.bad; stats[1] = t.max_abs; stats[2] = t.max_mul_in; stats[3] = t.max_wide_in; stats[4] = ck.worst;
        constexpr RegBounds ob = Out::in();
        for (int r = 0; r < 16; r++) { stats[5 + r] = sb.v[r]; stats[21 + r] = ob.v[r]; }
    }
}
// scalar checks of the two new field operations against exact arithmetic: returns the number of violations
int hm_check_mul_root4(const double* a, int count)
{
    int bad = 0;
    for (int i = 0; i < count; i++) {
        const double r = fpf::mul_root4(a[i]);
        const __int128 diff = (__int128)(int64_t)a[i] * (int64_t)fpf::ROOT4 - (__int128)(int64_t)r;
        if (diff % (__int128)P != 0) bad++;
        if (std::fabs(r) > fpf::AFTER_MUL_ROOT4 * fpf::P || r != std::nearbyint(r)) bad++;
    }
    return bad;
}
int hm_check_mulmod_add(const double* a, const double* w, const double* c, int count, int wide)
{
    int bad = 0;
    for (int i = 0; i < count; i++) {
        const double r = wide ? fpf::mulmod_add_wide(a[i], w[i], c[i]) : fpf::mulmod_add(a[i], w[i], c[i]);
        const __int128 diff = (__int128)(int64_t)a[i] * (int64_t)w[i] + (__int128)(int64_t)c[i] - (__int128)(int64_t)r;
        if (diff % (__int128)P != 0) bad++;
        const double bound = (wide ? 1.0 : 0.5) + fpf::GROW_ADD * std::fabs(a[i]) / fpf::P + 1e-9;
        if (std::fabs(r) > bound * fpf::P || r != std::nearbyint(r)) bad++;
    }
    return bad;
}

double hm_p(void) { return fpf::P; }
}
