// kernels_lvl2q.hip.h -- blind rotation over the N = 2048 ring (64-bit torus, BASELINE.json configs[4]) with FOUR waves per
// rotation, each owning one QUARTER of the ring and its six limb sums IN REGISTERS.  Same semantics as
// blind_rotate_lvl2_kernel (kernels_lvl2.hip.h: the reference's Accumulate / __BlindRotate__ templates instantiated at
// lvl02, include/gatebootstrapping_gpu.cuh:115-345), different machine mapping.
//
// Why.  The eight-wave kernel of kernels_lvl2.hip.h keeps its twelve half sums in LDS (96 KiB) and its eight waves in
// lock-step phases: LDS is full with ONE rotation per CU, the two waves of a SIMD always belong to the same rotation, and
// whatever one of them waits for (a barrier, an LDS round trip, the key) nobody covers: 75 % of the issue floor of its own
// instruction stream.  The wave-per-rotation kernel of the N = 1024 ring reaches 90 % because the two waves of a SIMD are
// INDEPENDENT.  This kernel gets the same property for N = 2048:
//
//   X^2048 + 1 = (X^512 - zeta)(X^512 + zeta)(X^512 - zeta^3)(X^512 + zeta^3)      zeta = 5440, zeta^4 = -1 (fpfield.h)
//
//   the first two forward stages applied to four gadget digits d0..d3 at e, e + 512, e + 1024, e + 1536 are four EXACT
//   linear forms (|value| < 2^46: three FMAs, no reduction)
//       q0 = d0 + I d2 + zeta d1 + zeta^3 d3        q1 = d0 + I d2 - zeta d1 - zeta^3 d3
//       q2 = d0 - I d2 + zeta^3 d1 + zeta d3        q3 = d0 - I d2 - zeta^3 d1 - zeta d3
//   and leave four independent 512-point transforms (ntt_wave512.h code, tables root_q[m + g] = root[4m + q m + g]).
//   Wave q of a 4-wave workgroup runs quarter q of all eight TRGSW rows one after the other and keeps the six sums
//   (output x limb) of its quarter -- 6 x 8 doubles per lane -- in registers: no LDS sums, no LDS atomics, no barrier
//   between rows; the six inverse quarter transforms follow straight out of the registers.  What crosses waves is small:
//   the gadget digits going in (16-bit words: 32 KiB) and the inverse-transformed quarters coming back for the last two
//   inverse stages (in six rounds of 16 KiB).  LDS per rotation: 70 KiB, so TWO workgroups share a CU and every SIMD
//   holds two waves of different rotations.
//
// Ownership.  Wave w owns, in registers, the accumulator words at e0 + 512 t (t = 0..3) for e0 = 128 w + lane + 64 s
// (s = 0, 1), both components: the four positions one radix-4 butterfly of the split couples.  Per CMux step:
//   (1) every wave leaves a copy of its words in LDS, barrier, reads the rotated operand, barrier (the copy is dead);
//   (2) decomposes its words, all four digits, and writes digit d of (e0, e0 + 512, e0 + 1024, e0 + 1536) as ONE 64-bit word
//       (four int16) into row (j, d) of the digit buffer, barrier;
//   (3) for each of the 8 rows: eight 64-bit reads, split, forward quarter transform, reduce, six products into the
//       register sums (key polynomials streamed from L2 two ahead);  barrier (the digit buffer is dead);
//   (4) for each of the 6 sums: inverse quarter transform, result to the exchange buffer (alternating halves of the same
//       32 KiB), barrier, every wave reads the four quarters at its e0, runs the last two inverse stages, lifts, shifts the
//       limb and adds into its accumulator words.
// Ten barriers of four waves per step; the partner workgroup on the CU runs through them.
#pragma once
#include "kernels_lvl2.hip.h"
#include "ntt_wave512.h"

namespace cufhe_amd {

constexpr int kQWaves = 4;
constexpr int kQThreads = 64 * kQWaves;
constexpr int kQPoints = 512;
static_assert(k2N == kQWaves * kQPoints && kQPoints == kH, "four quarter transforms of ntt_wave512.h");
// NTT-domain key of this kernel: [step][quarter][row][out * 3 + limb][c >> 1][lane][c & 1] doubles (c = register of layout C):
// the 48 polynomials a quarter wave multiplies with in one step are one contiguous 192 KiB block, in the order it uses them
constexpr size_t kQKeyPolyDoubles = kQPoints;
constexpr size_t kQKeyQuarterDoubles = (size_t)k2BkRows * k2Prods * kQPoints;      // 24576
static_assert(kQWaves * kQKeyQuarterDoubles == k2BkStepDoubles, "same bytes per step as the half-transform layout");
constexpr double kZeta3 = 160989184000.0;      // zeta^3 = 5440^3 (exact in a double)
static_assert(kZeta3 == fpf::ROOT8 * fpf::ROOT8 * fpf::ROOT8 && fpf::ROOT4 == fpf::ROOT8 * fpf::ROOT8, "roots of the split");
// exactness of the split on gadget digits: |q| <= (Bg/2) (1 + I + zeta + zeta^3) stays an exact double with room to spare
constexpr double kQSplitDigitBound = (double)(1u << (k2Bgbit - 1)) * (1.0 + fpf::ROOT4 + fpf::ROOT8 + kZeta3);
static_assert(kQSplitDigitBound < 9007199254740992.0 / 128.0, "split of the digits is not exact");
// ... and of the packed form the kernel evaluates it in: A + cI B + (c1 - 65536) d1 + (c3 - 65536 cI) d3 with |A|, |B| < 2^25, |d| <= 2^8
static_assert(33554432.0 * (1.0 + fpf::ROOT4) + 256.0 * (kZeta3 + 65536.0) + 256.0 * (kZeta3 + 65536.0 * fpf::ROOT4) < 9007199254740992.0 / 4.0,
              "packed split: a partial sum leaves the exact range");
static_assert(k2Bgbit <= 15, "a digit plus Bg/2 must fit the low 16-bit field");

// lazy-reduction schedule of the forward quarter transform below (units of p): nine stages, the last two wide, the addend of
// the last one reduced: the spectrum comes out below the narrow multiplication's limit without a reduction pass of its own
constexpr double quarter_forward_bound(double in)
{
    double b = in;
    for (int s = 0; s <= 6; s++) {
        if (b >= fpf::LIM_NARROW) return -1.0;
        b = b + fpf::after_mulmod(b);
    }
    if (b >= fpf::LIM_WIDE) return -1.0;
    b = b + fpf::after_mulmod_wide(b);                // stage 7
    if (b >= fpf::LIM_WIDE) return -1.0;
    return 0.5001 + fpf::after_mulmod_wide(b);         // stage 8: reduce(a) +- mulmod_wide(b, w)
}
static_assert(quarter_forward_bound(kQSplitDigitBound / fpf::P) > 0 && quarter_forward_bound(kQSplitDigitBound / fpf::P) < fpf::LIM_NARROW,
              "forward quarter transform of digits: the spectrum must be a legal operand of mulmod");
// products of that spectrum with key residues (<= p/2), eight rows into one register sum
static_assert(k2BkRows * fpf::after_mulmod(quarter_forward_bound(kQSplitDigitBound / fpf::P)) < fpf::LIM_WIDE, "register sums of eight rows");
// key limbs: |limb| <= 2^21, first split stage exact (2^21 (1 + I) = 2^46), second by a modular product: |in| <= 2^46 / p + 0.55
static_assert(quarter_forward_bound(0.09 + 0.56) > 0 && quarter_forward_bound(0.09 + 0.56) < fpf::LIM_WIDE, "forward quarter transform of key limbs");

// ---- LDS map of blind_rotate_lvl2q_kernel ----
constexpr int kQLdsR = 0;                                             // 32 KiB: accumulator copy | digit buffer | exchange buffer
constexpr int kQLdsRBytes = 2 * k2N * 8;                              // 32768
constexpr int kQLdsTiles = kQLdsR + kQLdsRBytes;                      // 4 transpose tiles of a 512-point transform
constexpr int kQLdsTab = kQLdsTiles + kQWaves * kTile512Bytes;        // per quarter [tb_fwd 56 | tb_inv 56 | tc_fwd 448] doubles
constexpr int kQTabDoubles = 2 * 7 * 8 + 7 * 64;                      // 560
constexpr int kQTabBytes = kQTabDoubles * 8;                          // 4480
constexpr int kQLdsAbar = kQLdsTab + kQWaves * kQTabBytes;
constexpr int kQLdsR4 = kQLdsAbar + kAbarBytes + 16;                  // per quarter [uwb_fwd 8 | uwb_inv 8 | uwc_fwd 64 | uwc_inv 64] doubles
constexpr int kQR4Doubles = 144;                                      // Ntt512Tables: uwb_fwd 8 | uwb_inv 8 | uwc_fwd 64 | uwc_inv 64
constexpr int kQLdsBytes = kQLdsR4 + kQWaves * kQR4Doubles * 8;       // 74768
static_assert(2 * kQLdsBytes <= 160 * 1024, "two rotations per CU");
static_assert(8 * k2BkRows * kQPoints == kQLdsRBytes && 2 * kQWaves * kQPoints * 8 == kQLdsRBytes, "digit buffer and exchange halves fit the accumulator copy's region");

__device__ __forceinline__ void ct_three_stages_w2(double (&x)[kRegs8], const double (&tw)[7])
{
    const double w0 = tw[0];
#pragma unroll
    for (int r = 0; r < 4; r++) ct_bfly<false>(x[r], x[r + 4], w0);
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const double w = tw[1 + g];
#pragma unroll
        for (int r = 0; r < 2; r++) ct_bfly<true>(x[4 * g + r], x[4 * g + r + 2], w);
    }
    // last stage: only the addend is reduced (2 operations per butterfly instead of a reduction of both outputs afterwards):
    // |out| <= 0.5 + (1 + 0.0973 |b|) p
#pragma unroll
    for (int g = 0; g < 4; g++) {
        x[2 * g] = fpf::reduce(x[2 * g]);
        ct_bfly<true>(x[2 * g], x[2 * g + 1], tw[3 + g]);
    }
}

// ---- the radix-4 schedules of the quarter transforms (machinery: ntt_wave512.h, q4) ----
namespace q4 {
// forward schedule on gadget digits after the exact split (|in| <= kQSplitDigitBound / p < 0.05)
constexpr int kInMicro = 50000;
static_assert(kQSplitDigitBound / fpf::P < kInMicro * 1e-6, "split digits exceed the schedule's input bound");
using FA0 = U8<kInMicro>;
using FA1 = CtR4<FA0>;
using FB0 = Xpose8<CtC<FA1, false>>;
using FB1 = CtR4<FB0>;
using FC0 = Xpose8<CtC<FB1, false>>;
using FC1 = CtR4<FC0>;
using FSpec = CtC<FC1, true>;
static_assert(max8(FSpec::in()) < fpf::LIM_NARROW, "radix-4 forward quarter transform: the spectrum must be a legal operand of mulmod");
static_assert(k2BkRows * fpf::after_mulmod(max8(FSpec::in())) < fpf::LIM_WIDE, "register sums of eight rows (radix-4 forward)");
// inverse schedule on reduced sums; registers above 2^51 (2.57 p) are reduced between passes
constexpr int kLim = 2570;
using IC0 = U8<500100>;
using IC1 = Red8<GsC<IC0>, kLim>;
using IC2 = Red8<GsR4<IC1>, kLim>;
using IB0 = Xpose8<IC2>;
using IB1 = Red8<GsC<IB0>, kLim>;
using IB2 = Red8<GsR4<IB1>, kLim>;
using IA0 = Xpose8<IB2>;
using IA1 = Red8<GsC<IA0>, kLim>;
using IOut = Red8<GsR4<IA1>, kLim>;
// the coefficient-wise tail adds four of these and multiplies differences of two: 4 x 2.57 < 10.285, 2 x 2.57 < 5.142
static_assert(max8(IOut::in()) <= kLim * 0.001 && 4 * max8(IOut::in()) < fpf::LIM_WIDE && 2 * max8(IOut::in()) < fpf::LIM_NARROW,
              "radix-4 inverse quarter transform: outputs exceed what the last two inverse stages accept");
}  // namespace q4

// per-lane addresses of a quarter wave
struct QuarterCtx {
    char *a1, *b1, *b2;               // the wave's transpose tile (slot maps of ntt_wave512.h; c2 == a1)
    const char* tb_fwd;               // LDS tables of this quarter + 8 lam      (+ 64 k); tb_inv = tb_fwd + 8 * 56
    const char* tc_fwd;               // ... + 8 lane                            (+ 512 k)
    const char* tc_mirror;            // forward stage 6-8 twiddles of quarter 3 - q at lane 63 - L: the inverse ones, negated and re-indexed
    const Ntt512Tables* gt;           // global tables of this quarter (wave-uniform stage 0-2 twiddles; slot 7: their product u w / v w)
    const char* r4b;                  // LDS copy of this quarter's Ntt512Tables::uwb_fwd .. uwc_inv + 8 lam: uwb_fwd (+ 64: uwb_inv)
    const char* r4c;                  // ... + 8 * 16 + 8 lane: uwc_fwd (+ 512: uwc_inv)
};

// forward quarter transform: x in layout A (natural: e = lane + 64 reg), out in layout C; |in| <= 0.65 p, |out| < 10.285 p
__device__ __forceinline__ void quarter_forward(double (&x)[kRegs8], const QuarterCtx& c)
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
    CUFHE_AMD_XPOSE8(c.b2, 64, c.a1, 8 * 72)          // B -> C
    ct_three_stages_w2(x, twc);
}
// inverse quarter transform: x in layout C with |x| <= p/2 (+ a tie), out in layout A with |x| <= p, not scaled.  Its stage 8-6
// twiddles are not stored: root^-1[M + G] = -root[M + (M - 1 - G)] (root[i] = psi^bitrev(i), psi^2048 = -1) reads, for the quarter
// tables, inv_q[m + g] = -fwd_{3-q}[m + (m - 1 - g)]: lane L takes the forward twiddle of quarter 3 - q at lane 63 - L, index k -> 3 (2^lvl - 1) - k
__device__ __forceinline__ void quarter_inverse(double (&x)[kRegs8], const QuarterCtx& c)
{
    double twc[7];
    twc[0] = -lds_ld(c.tc_mirror, 0);
    twc[1] = -lds_ld(c.tc_mirror, 512 * 2);
    twc[2] = -lds_ld(c.tc_mirror, 512 * 1);
#pragma unroll
    for (int k = 3; k < 7; k++) twc[k] = -lds_ld(c.tc_mirror, 512 * (9 - k));
    double twb[7];
#pragma unroll
    for (int k = 0; k < 7; k++) twb[k] = lds_ld(c.tb_fwd, 8 * 56 + 64 * k);
    gs_three_stages<-1>(x, TwArr{twc});               // s8 s7 s6: .5 -> 4
    CUFHE_AMD_XPOSE8(c.a1, 8 * 72, c.b2, 64)          // C -> B
    gs_three_stages<0>(x, TwArr{twb});                // s5 (wide, reduce) s4 s3: -> 2
    CUFHE_AMD_XPOSE8(c.b1, 64, c.a1, 8 * 68)          // B -> A
    gs_three_stages<1>(x, TwUniform{c.gt->tu_inv});   // s2 s1 (wide, reduce) s0: -> 1
}

// The same two transforms in radix-4 form (q4 above): identical values mod p, so identical words.
__device__ __forceinline__ void quarter_forward_r4(double (&x)[kRegs8], const QuarterCtx& c)
{
    const double* tu = c.gt->tu_fwd;
    q4::ct_r4_pass<q4::FA0>(x, tu[0], tu[1], tu[7]);
    q4::ct_c_stage<q4::FA1, false>(x, TwUniform{tu});
    double twb[7];
    twb[0] = lds_ld(c.tb_fwd, 0); twb[1] = lds_ld(c.tb_fwd, 64); twb[2] = lds_ld(c.r4b, 0);
#pragma unroll
    for (int k = 3; k < 7; k++) twb[k] = lds_ld(c.tb_fwd, 64 * k);
    CUFHE_AMD_XPOSE8(c.a1, 8 * 68, c.b1, 64)          // A -> B
    q4::ct_r4_pass<q4::FB0>(x, twb[0], twb[1], twb[2]);
    q4::ct_c_stage<q4::FB1, false>(x, TwArr{twb});
    double twc[7];
    twc[0] = lds_ld(c.tc_fwd, 0); twc[1] = lds_ld(c.tc_fwd, 512); twc[2] = lds_ld(c.r4c, 0);
#pragma unroll
    for (int k = 3; k < 7; k++) twc[k] = lds_ld(c.tc_fwd, 512 * k);
    CUFHE_AMD_XPOSE8(c.b2, 64, c.a1, 8 * 72)          // B -> C
    q4::ct_r4_pass<q4::FC0>(x, twc[0], twc[1], twc[2]);
    q4::ct_c_stage<q4::FC1, true>(x, TwArr{twc});
}
__device__ __forceinline__ void quarter_inverse_r4(double (&x)[kRegs8], const QuarterCtx& c)
{
    double twc[7];
    twc[0] = -lds_ld(c.tc_mirror, 0);
    twc[1] = -lds_ld(c.tc_mirror, 512 * 2);
    twc[2] = lds_ld(c.r4c, 512);                      // v w
#pragma unroll
    for (int k = 3; k < 7; k++) twc[k] = -lds_ld(c.tc_mirror, 512 * (9 - k));
    double twb[7];
    twb[0] = lds_ld(c.tb_fwd, 8 * 56); twb[1] = lds_ld(c.tb_fwd, 8 * 56 + 64); twb[2] = lds_ld(c.r4b, 64);
#pragma unroll
    for (int k = 3; k < 7; k++) twb[k] = lds_ld(c.tb_fwd, 8 * 56 + 64 * k);
    q4::gs_c_stage<q4::IC0>(x, TwArr{twc});                    // s8
    q4::reduce_above8<q4::GsC<q4::IC0>, 0, q4::kLim>(x);
    q4::gs_r4_pass<q4::IC1>(x, twc[0], twc[1], twc[2]);        // s7 s6
    q4::reduce_above8<q4::GsR4<q4::IC1>, 0, q4::kLim>(x);
    CUFHE_AMD_XPOSE8(c.a1, 8 * 72, c.b2, 64)          // C -> B
    q4::gs_c_stage<q4::IB0>(x, TwArr{twb});                    // s5
    q4::reduce_above8<q4::GsC<q4::IB0>, 0, q4::kLim>(x);
    q4::gs_r4_pass<q4::IB1>(x, twb[0], twb[1], twb[2]);        // s4 s3
    q4::reduce_above8<q4::GsR4<q4::IB1>, 0, q4::kLim>(x);
    CUFHE_AMD_XPOSE8(c.b1, 64, c.a1, 8 * 68)          // B -> A
    const double* tu = c.gt->tu_inv;
    q4::gs_c_stage<q4::IA0>(x, TwUniform{tu});                 // s2
    q4::reduce_above8<q4::GsC<q4::IA0>, 0, q4::kLim>(x);
    q4::gs_r4_pass<q4::IA1>(x, tu[0], tu[1], tu[7]);           // s1 s0
    q4::reduce_above8<q4::GsR4<q4::IA1>, 0, q4::kLim>(x);
}
#ifdef CUFHE_AMD_Q_RADIX2        // experiment (tools/build_variant.py): the radix-2 quarter transforms (profiles/r05_lvl2_ab.txt)
constexpr bool kQuarterR4 = false;
#else
constexpr bool kQuarterR4 = true;
#endif

__device__ __forceinline__ void load_key8(double2 (&b)[4], const double* poly, int lane)
{
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_NO_BK)       // timing-only: no key loads
#pragma unroll
    for (int q = 0; q < 4; q++) b[q] = make_double2(1234567.0 + q + (double)(uintptr_t)poly * 1e-30, 7654321.0 + lane);
    return;
#endif
    const double2* p = (const double2*)poly;
#pragma unroll
    for (int q = 0; q < 4; q++) b[q] = p[q * 64 + lane];
}

// ----------------------------------------------------------------------------------
// BK (torus, uint64) -> NTT domain in the quarter layout.  One wave per (polynomial, limb): split into limbs as
// bk2_to_ntt_kernel, the two split stages (the first exact, the second by modular products), four quarter transforms,
// scaled by 2048^-1.  tq: the four quarter tables in global memory.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kNttThreads) void bk2q_to_ntt_kernel(
    double* __restrict__ bk_ntt, const uint64_t* __restrict__ bk, size_t polys,
    const Ntt512Tables* __restrict__ tq, double n_inverse)
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
    const int lo = lane & 7, hi = lane >> 3;
    const int tile_off = wave * kTile512Bytes;
#pragma unroll 1
    for (int q = 0; q < 4; q++) {
        double x[kRegs8];
#pragma unroll
        for (int c = 0; c < kRegs8; c++) {
            double v[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                int64_t s = (int64_t)src[lane + 64 * c + kQPoints * t];
                int64_t l = (s << (64 - k2LimbBits)) >> (64 - k2LimbBits);
                for (int m = 0; m < limb; m++) {
                    s = (s - l) >> k2LimbBits;
                    l = (m + 1 == k2Limbs - 1) ? s : (s << (64 - k2LimbBits)) >> (64 - k2LimbBits);
                }
                v[t] = (double)l;
            }
            const double sgn = (q & 2) ? -1.0 : 1.0;
            const double u = __builtin_fma(sgn * fpf::ROOT4, v[2], v[0]);          // exact: |I limb| < 2^46
            const double u2 = __builtin_fma(sgn * fpf::ROOT4, v[3], v[1]);
            const double t2 = fpf::mulmod(u2, (q & 2) ? kZeta3 : fpf::ROOT8);
            x[c] = (q & 1) ? u - t2 : u + t2;
        }
        QuarterCtx ctx;
        ctx.a1 = smem + opaque(tile_off + 8 * lane);
        ctx.b1 = smem + opaque(tile_off + 8 * (68 * lo + hi));
        ctx.b2 = smem + opaque(tile_off + 8 * (lo + 72 * hi));
        ctx.tb_fwd = (const char*)tq[q].tb_fwd + 8 * lo;
        ctx.tc_fwd = (const char*)tq[q].tc_fwd + 8 * lane;
        ctx.tc_mirror = nullptr;
        ctx.gt = tq + q;
        quarter_forward(x, ctx);
        double2* dst = (double2*)(bk_ntt + ((((step * kQWaves + q) * k2BkRows + row) * k2Prods) + out * k2Limbs + limb) * kQKeyPolyDoubles);
#pragma unroll
        for (int c2 = 0; c2 < 4; c2++) {
            double2 v;
            v.x = fpf::reduce(fpf::mulmod_wide(x[2 * c2], n_inverse));
            v.y = fpf::reduce(fpf::mulmod_wide(x[2 * c2 + 1], n_inverse));
            dst[c2 * 64 + lane] = v;
        }
    }
}

// ----------------------------------------------------------------------------------
// Blind rotate lvl02 + sample extract: one 4-wave workgroup per rotation, two workgroups per CU.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kQThreads, 2) void blind_rotate_lvl2q_kernel(
    const RotDesc2* __restrict__ descs, int count, const double* __restrict__ bk_ntt,
    const Ntt512Tables* __restrict__ tq, int steps, uint64_t* __restrict__ acc_dump)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = blockIdx.x;
    if (g >= count) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // = the quarter this wave transforms
    const int lane = tid & 63;
    uint16_t* abar_lds = (uint16_t*)(smem + kQLdsAbar);
    uint32_t* bbar_slot = (uint32_t*)(smem + kQLdsAbar + kAbarBytes);

    const RotDesc2 d = descs[g];
    for (int i = tid; i <= kLvl0N; i += kQThreads) {
        const uint32_t c = (uint32_t)d.ca * d.in0[i] + (uint32_t)d.cb * d.in1[i];
        if (i < kLvl0N) abar_lds[i] = (uint16_t)((c + (1u << (32 - 2 - k2Nbit))) >> (32 - 1 - k2Nbit));
        else *bbar_slot = 2 * k2N - ((c + d.off) >> (32 - 1 - k2Nbit));
    }
    for (int i = tid; i < kQWaves * kQTabDoubles; i += kQThreads) {
        const int q = i / kQTabDoubles, k = i % kQTabDoubles;
        // [tb_fwd 56 | tb_inv 56] are contiguous in Ntt512Tables, tc_fwd follows
        ((double*)(smem + kQLdsTab))[i] = k < 112 ? tq[q].tb_fwd[k] : tq[q].tc_fwd[k - 112];
    }
    for (int i = tid; i < kQWaves * kQR4Doubles; i += kQThreads) ((double*)(smem + kQLdsR4))[i] = tq[i / kQR4Doubles].uwb_fwd[i % kQR4Doubles];
    __syncthreads();

    QuarterCtx ctx;
    {
        const int lo = lane & 7, hi = lane >> 3;
        const int tile_off = kQLdsTiles + wave * kTile512Bytes;
        ctx.a1 = smem + opaque(tile_off + 8 * lane);
        ctx.b1 = smem + opaque(tile_off + 8 * (68 * lo + hi));
        ctx.b2 = smem + opaque(tile_off + 8 * (lo + 72 * hi));
        ctx.tb_fwd = smem + opaque(kQLdsTab + wave * kQTabBytes + 8 * lo);
        ctx.tc_fwd = smem + opaque(kQLdsTab + wave * kQTabBytes + 8 * 112 + 8 * lane);
        ctx.tc_mirror = smem + opaque(kQLdsTab + (3 - wave) * kQTabBytes + 8 * 112 + 8 * (63 - lane));
        ctx.gt = tq + wave;
        ctx.r4b = smem + opaque(kQLdsR4 + wave * kQR4Doubles * 8 + 8 * lo);
        ctx.r4c = smem + opaque(kQLdsR4 + wave * kQR4Doubles * 8 + 8 * 16 + 8 * lane);
    }
    // the split's constants of this quarter: q = d0 + cI d2 + c1 d1 + c3 d3
    const double cI = (wave & 2) ? -fpf::ROOT4 : fpf::ROOT4;
    const double sg = (wave & 1) ? -1.0 : 1.0;
    const double c1 = sg * ((wave & 2) ? kZeta3 : fpf::ROOT8);
    const double c3 = sg * ((wave & 2) ? fpf::ROOT8 : kZeta3);
    const double k1 = c1 - 65536.0, k3 = c3 - 65536.0 * cI, kOff = (double)(1u << (k2Bgbit - 1)) * (1.0 + cI);

    // accumulator words of this lane: component j, s (e0 = 128 wave + lane + 64 s), t (position e0 + 512 t)
    const int e_base = 128 * wave + lane;
    uint64_t acc[2][2][4];
    {
        const uint32_t bbar = *bbar_slot;     // RotatedTestVector<lvl2param>, include/gatebootstrapping_gpu.cuh:29-52
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t e = (uint32_t)(e_base + 64 * s + kQPoints * t);
                const bool neg = (bbar != 2 * k2N) && ((e < (bbar & (k2N - 1))) != ((bbar >> k2Nbit) != 0));
                acc[0][s][t] = 0ull;
                acc[1][s][t] = neg ? 0ull - k2Mu : k2Mu;
            }
    }
    char* own = smem + opaque(kQLdsR + 8 * e_base);        // + 16384 j + 512 s + 4096 t: this lane's words in the copy / its exchange slots
    auto publish_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int t = 0; t < 4; t++) *(uint64_t*)(own + 16384 * j + 512 * s + 4096 * t) = acc[j][s][t];
    };

    double sums[k2Prods][kRegs8];
#ifdef CUFHE_AMD_Q_PF3
    constexpr int kKb = 4;              // key buffers: loads run kKb - 1 polynomials ahead of their use
#else
    constexpr int kKb = 3;
#endif
    constexpr int kRowUnroll = (kKb == 3) ? 1 : 2;      // (rows per loop body) * 6 polynomials must be a multiple of kKb
    static_assert((kRowUnroll * k2Prods) % kKb == 0 && k2BkRows % kRowUnroll == 0, "key buffer rotation");
    double2 kb[kKb][4];
    const double* key_q = bk_ntt + (size_t)wave * kQKeyQuarterDoubles;
    auto step_key = [&](int step) {
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_BK0)
        return key_q + (size_t)(step & 1) * k2BkStepDoubles;
#else
        return key_q + (size_t)step * k2BkStepDoubles;
#endif
    };
    if (steps > 0) {
#pragma unroll
        for (int b = 0; b < kKb - 1; b++) load_key8(kb[b], step_key(0) + (size_t)b * kQKeyPolyDoubles, lane);
    }
    publish_acc();
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_PHASES)
    // timing-only: cycles of this wave per phase: [0] barrier (copies visible), [1] rotated reads + barrier, [2] digits + barrier,
    // [3] split + forward, [4] products, [5] barrier (digit buffer dead), [6] inverse transforms, [7] exchange barriers,
    // [8] last stages + lift + accumulate, [9] publish
    unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#define CUFHE_AMD_PHASEQ(k) { const unsigned long long tn = __builtin_readcyclecounter(); ph[k] += tn - tc; tc = tn; }
#else
#define CUFHE_AMD_PHASEQ(k)
#endif
#pragma unroll 1
    for (int i = 0; i < steps; i++) {
        __syncthreads();                                      // every owner's copy of the accumulator is in LDS
        CUFHE_AMD_PHASEQ(0)
        const uint32_t abar = __builtin_amdgcn_readfirstlane((uint32_t)abar_lds[i]);
        const int alo = (int)(abar & (k2N - 1));
        const bool ahi = (abar >> k2Nbit) != 0;
        const double* key = step_key(i);
        // L2 warming: the 64 workgroups resident on an XCD (blocks b, b + 8, ...: round-robin dispatch -- for speed only) walk the
        // key together, and whoever touches a line first waits for the fabric.  Each workgroup touches 1/64 of the key of step
        // i + 2 (768 KiB / 64 = 96 lines of 128 bytes: 24 lanes of each wave, one 4-byte load), so that the XCD's L2 holds a step
        // before anyone needs it.  Nothing waits for these loads: they are "used" after the row loop.
        uint32_t warm = 0;
#ifdef CUFHE_AMD_Q_WARM
        if (lane < 24 && i + 2 < steps)
            warm = *(const uint32_t*)((const char*)(bk_ntt + (size_t)(i + 2) * k2BkStepDoubles) + (size_t)((blockIdx.x >> 3) & 63) * 12288 + (wave * 24 + lane) * 128);
#endif
        // (1) the rotated operand (X^abar acc_j) at this lane's positions
        uint64_t rot[2][2][4];
        {
            const int rb = (e_base - alo) & (k2N - 1);
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int idx = (rb + 64 * s + kQPoints * t) & (k2N - 1);
                    rot[0][s][t] = *(const uint64_t*)(smem + kQLdsR + 8 * idx);
                    rot[1][s][t] = *(const uint64_t*)(smem + kQLdsR + 16384 + 8 * idx);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                      // all rotated reads done: the copy is dead
        CUFHE_AMD_PHASEQ(1)
        // (2) decomposition: digit d of the four positions of (j, s) as one 64-bit word of row (j, d)
        {
            char* dig_out = smem + opaque(kQLdsR + 8 * e_base);          // + 4096 row + 512 s
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    // positions t = 1, 3 (the high 16-bit fields) carry the signed digit (sign mask applied: two's complement),
                    // positions t = 0, 2 (the low fields) the digit + Bg/2 as the decomposition produces it (no sign mask)
                    uint64_t tmp[4];
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const int e = e_base + 64 * s + kQPoints * t;
                        const bool neg = (e < alo) != ahi;
                        tmp[t] = (neg ? 0ull - rot[j][s][t] : rot[j][s][t]) - acc[j][s][t] + decomp_offset2();
                        if (t & 1) tmp[t] ^= decomp_signmask2();
                    }
#pragma unroll
                    for (int dd = 0; dd < k2L; dd++) {
                        constexpr int kTop = 64 - k2Bgbit;
                        const int pos = kTop - k2Bgbit * dd;
                        uint32_t dg[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const uint32_t f = pos >= 32 ? (uint32_t)(tmp[t] >> 32) >> (pos - 32)
                                                         : __builtin_amdgcn_alignbit((uint32_t)(tmp[t] >> 32), (uint32_t)tmp[t], (uint32_t)pos);
                            dg[t] = (t & 1) ? (uint32_t)__builtin_amdgcn_sbfe(f, 0u, (uint32_t)k2Bgbit) : (f & ((1u << k2Bgbit) - 1));
                        }
                        uint2 w;
                        w.x = dg[0] | (dg[1] << 16);      // d0 + 256 in [0, 512) | d1 << 16: as a signed word 65536 d1 + d0 + 256
                        w.y = dg[2] | (dg[3] << 16);
                        *(uint2*)(dig_out + 4096 * (j * k2L + dd) + 512 * s) = w;
                    }
                }
        }
        __syncthreads();                                      // the digit buffer is complete
        CUFHE_AMD_PHASEQ(2)
        // (3) the eight rows: split, forward quarter transform, six products into the register sums
#pragma unroll
        for (int p = 0; p < k2Prods; p++)
#pragma unroll
            for (int c = 0; c < kRegs8; c++) sums[p][c] = 0.0;
        const char* dig_in = smem + opaque(kQLdsR + 8 * lane);
#pragma unroll 1
        for (int row0 = 0; row0 < k2BkRows; row0 += kRowUnroll) {
#pragma unroll
            for (int rr = 0; rr < kRowUnroll; rr++) {
                const int row = row0 + rr;
                double x[kRegs8];
                {
                    uint2 w[kRegs8];
#pragma unroll
                    for (int c = 0; c < kRegs8; c++) w[c] = *(const uint2*)(dig_in + 4096 * row + 512 * c);
#pragma unroll
                    for (int c = 0; c < kRegs8; c++) {
                        // A = 65536 d1 + d0 + 256, B = 65536 d3 + d2 + 256 (one conversion each), d1, d3 from the high fields:
                        // d0 + cI d2 + c1 d1 + c3 d3 = A + cI B + (c1 - 65536) d1 + (c3 - 65536 cI) d3 - 256 (1 + cI): ten operations, all exact
                        const double A = (double)(int)w[c].x, d1 = (double)((int)w[c].x >> 16);
                        const double B = (double)(int)w[c].y, d3 = (double)((int)w[c].y >> 16);
                        x[c] = __builtin_fma(k3, d3, __builtin_fma(k1, d1, __builtin_fma(cI, B, A))) - kOff;      // exact: every partial sum < 2^51
                    }
                }
                if constexpr (kQuarterR4) quarter_forward_r4(x, ctx);
                else quarter_forward(x, ctx);                  // |x| <= 2.5 p: a legal operand of mulmod as it is
                CUFHE_AMD_PHASEQ(3)
                const double* krow = key + (size_t)row * (k2Prods * kQKeyPolyDoubles);
#pragma unroll
                for (int p = 0; p < k2Prods; p++) {
                    // kKb - 1 polynomials ahead; past the last row of the step the stream continues with the next step's block
                    const int ahead = row * k2Prods + p + (kKb - 1);
                    const double* nxt = (ahead < k2BkRows * k2Prods) ? krow + (size_t)(p + kKb - 1) * kQKeyPolyDoubles
                                                                      : step_key(i + 1 < steps ? i + 1 : i) + (size_t)(ahead - k2BkRows * k2Prods) * kQKeyPolyDoubles;
                    load_key8(kb[(rr * k2Prods + p + kKb - 1) % kKb], nxt, lane);
                    const double2(&b)[4] = kb[(rr * k2Prods + p) % kKb];
#pragma unroll
                    for (int c2 = 0; c2 < 4; c2++) {
                        sums[p][2 * c2] += fpf::mulmod(x[2 * c2], b[c2].x);
                        sums[p][2 * c2 + 1] += fpf::mulmod(x[2 * c2 + 1], b[c2].y);
                    }
                }
                CUFHE_AMD_PHASEQ(4)
            }
        }
        asm volatile("" :: "v"(warm));                        // the warming load ends here, long after its issue
        __syncthreads();                                      // every wave has read its last digits: the buffer is dead
        CUFHE_AMD_PHASEQ(5)
        // (4) six rounds: inverse quarter transform of sum (out o, limb l), exchange, last two inverse stages at this lane's
        // positions, lift, shift, add.  Round k uses half k & 1 of the region; its slots [qq][e0] are the accumulator copy's
        // [j = k & 1][e0 + 512 qq], so the copy written after the last round only overwrites slots this lane itself read.
        // The inverse transform of round k + 1 runs BEFORE the barrier of round k: the write of round k, the barrier and the reads
        // of round k have an inverse transform of independent work in between instead of an exposed LDS round trip each.  Round
        // k + 1 is written after the barrier of round k into the half round k - 1 used: every wave read that before it arrived.
        auto inverse_of = [&](double (&y)[kRegs8], int k) {
            const int p = (k & 1) * k2Limbs + (k >> 1);
#pragma unroll
            for (int c = 0; c < kRegs8; c++) y[c] = fpf::reduce(sums[p][c]);
            if constexpr (kQuarterR4) quarter_inverse_r4(y, ctx);
            else quarter_inverse(y, ctx);
        };
        auto exchange_write = [&](const double (&y)[kRegs8], int k) {
            char* ex = smem + opaque(kQLdsR + 16384 * (k & 1) + 4096 * wave + 8 * lane);
#pragma unroll
            for (int c = 0; c < kRegs8; c++) *(double*)(ex + 512 * c) = y[c];
        };
        double y[kRegs8];
        inverse_of(y, 0);
        exchange_write(y, 0);
        CUFHE_AMD_PHASEQ(6)
#pragma unroll
        for (int k = 0; k < k2Prods; k++) {
            const int l = k >> 1, o = k & 1;
            if (k + 1 < k2Prods) inverse_of(y, k + 1);
            CUFHE_AMD_PHASEQ(6)
            __syncthreads();
            CUFHE_AMD_PHASEQ(7)
            double v[2][4];
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int qq = 0; qq < 4; qq++) v[s][qq] = *(const double*)(own + 16384 * (k & 1) + 512 * s + 4096 * qq);
            if (k + 1 < k2Prods) exchange_write(y, k + 1);
#pragma unroll
            for (int s = 0; s < 2; s++) {
                // inverse of the split: (q0, q1) -> (h0, h0'), (q2, q3) -> (h1, h1'), then the halves; zeta^-1 = -zeta^3, zeta^-3 = -zeta, I^-1 = -I
                const double s01 = v[s][0] + v[s][1], d01 = fpf::mulmod(v[s][0] - v[s][1], -kZeta3);
                const double s23 = v[s][2] + v[s][3], d23 = fpf::mulmod(v[s][2] - v[s][3], -fpf::ROOT8);
                double r[4];
                r[0] = fpf::reduce(s01 + s23);
                r[2] = fpf::reduce(fpf::mulmod(s01 - s23, -fpf::ROOT4));
                r[1] = fpf::reduce(d01 + d23);
                r[3] = fpf::reduce(fpf::mulmod(d01 - d23, -fpf::ROOT4));
                // lift: the mantissa of r + 1.5 * 2^52 holds 2^51 + r, i.e. its bit pattern is 0x4338000000000000 + r as a 64-bit
                // integer; the constant, shifted like the limbs, is taken off once per word and step instead of once per limb
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const double tt = r[t] + fpf::MAGIC0;
                    uint64_t bits;
                    __builtin_memcpy(&bits, &tt, 8);
                    constexpr uint64_t kMagicBits = 0x4338000000000000ull;
                    constexpr uint64_t kAll = kMagicBits + (kMagicBits << k2LimbBits) + (kMagicBits << (2 * k2LimbBits));
                    acc[o][s][t] += bits << (k2LimbBits * l);
                    if (l == 0) acc[o][s][t] -= kAll;
                }
            }
            CUFHE_AMD_PHASEQ(8)
        }
        publish_acc();
        CUFHE_AMD_PHASEQ(9)
    }

    if (acc_dump) {
        uint64_t* o = acc_dump + (size_t)g * 2 * k2N + e_base;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int t = 0; t < 4; t++) o[j * k2N + 64 * s + kQPoints * t] = acc[j][s][t];
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
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int e = e_base + 64 * s + kQPoints * t;
                if (e == 0) {
                    o[0] = acc[0][s][t];
                    o[k2N] = acc[1][s][t];
                } else {
                    o[k2N - e] = 0ull - acc[0][s][t];
                }
            }
    }
#undef CUFHE_AMD_PHASEQ
}

}  // namespace cufhe_amd
