// fpfield.h -- exact arithmetic modulo a 50-bit prime carried in FP64 registers.
//
// Why FP64: on gfx950 a 64-bit integer modular butterfly costs 73-129 SIMD cycles
// per wave (Goldilocks shift / general multiply), two 31-bit Shoup butterflies 89,
// while the FP64 butterfly below costs 38 (tools/ubench/ubench_arith.hip, numbers in
// profiles/r01_ubench_arith.txt): v_fma_f64 is the widest exact multiplier the CU
// has at full VALU rate.  The external product is exact integer arithmetic
// (SURVEY.md F5), so any prime p with p/2 > max|sum| gives the reference's words.
//
// Bound: digits are in [-Bg/2, Bg/2) and the bootstrapping key is taken as SIGNED
// 32-bit torus words (congruent mod 2^32 to the reference's unsigned reading,
// include/ntt_gpu/ntt_gpuntt.cuh:495-496, and the result is only used mod 2^32), so
// |sum| <= (k+1) l N (Bg/2) 2^31 = 6144 * 32 * 2^31 = 2^48.585 < p/2 = 2^48.638.
//
// The prime has the form p = zeta^4 + 1 (zeta = 5440): zeta is then a primitive 8th root
// of unity of 13 bits and I = zeta^2 a 4th root of 25 bits, and psi (the 2048-th root) is
// chosen with psi^256 = zeta.  The twiddles of the first forward stage (I) and of half of
// the second (zeta) are so small that their products with gadget digits are exact in a
// double without any reduction: 3 FP64 operations per butterfly instead of 8.
//
// Representation: a residue is ANY integer-valued double x with |x| < 2^53 that is
// congruent to the value mod p ("lazy, balanced").  Additions never reduce; the
// multiplications below reduce as a side effect.  Every intermediate is an exact
// integer, so results do not depend on evaluation order.  In units of p:
//   2^53 / p = 10.285   (any value, additive headroom)
//   2^52 / p =  5.142   (largest |a| mulmod() accepts)
//   p / 2^53 = 0.09723  (growth of a multiplication result per unit of input)
#pragma once
#include <stdint.h>

// Ablation switches (CUFHE_AMD_ABL_*) exist for timing experiments only and make the kernels
// compute WRONG words.  They are honoured only in a diagnostic build, which cufhe_amd/build.py
// --diagnostic writes to libcufhe_amd_diag.so so that it can never be mistaken for the product.
#if (defined(CUFHE_AMD_ABL_NO_TW) || defined(CUFHE_AMD_ABL_NO_XPOSE) || defined(CUFHE_AMD_ABL_NO_BK) || \
     defined(CUFHE_AMD_ABL_BK0) || defined(CUFHE_AMD_ABL_NO_SYNC) || defined(CUFHE_AMD_ABL_PHASES) || defined(CUFHE_AMD_ABL_LL2_TIMEOUT) || defined(CUFHE_AMD_ABL_L2_NOATOM)) && !defined(CUFHE_AMD_DIAGNOSTIC_BUILD)
#error "CUFHE_AMD_ABL_* switches produce wrong results: use `python cufhe_amd/build.py --diagnostic=NAME[,NAME]` (defines CUFHE_AMD_DIAGNOSTIC_BUILD, output libcufhe_amd_diag.so)"
#endif

#if defined(__HIPCC__)
#define FPF_HD __host__ __device__ __forceinline__
#else
#define FPF_HD static inline
#endif

namespace fpf {

constexpr uint64_t ZETA8 = 5440ull;               // primitive 8th root of unity
constexpr uint64_t P_U64 = 875781160960001ull;    // zeta^4 + 1, prime, = 1 (mod 2^12), 2^49.64
constexpr uint64_t PSI_2048 = 423584205157050ull; // primitive 2048-th root of unity with psi^256 = zeta
constexpr double P = 875781160960001.0;
constexpr double PINV = 0x1.491cc17c934a8p-50;    // fl(1/p)
constexpr double ROOT4 = 29593600.0;              // I = zeta^2 = psi^512
constexpr double ROOT8 = 5440.0;                  // zeta = psi^256
constexpr double INV_ROOT4 = 0x1.22436485a6c7fp-25;   // fl(1/I)
constexpr double MAGIC0 = 6755399441055744.0;     // 1.5 * 2^52: adding it rounds to an integer
constexpr double MAGIC1 = 13510798882111488.0;    // 1.5 * 2^53: rounds to an even integer

// a*w mod p for |a| < 2^52 (5.142 p), |w| <= p/2.  |result| <= (0.5 + 0.0973 |a|/p) p.
FPF_HD double mulmod(double a, double w)
{
    const double h = a * w;
    const double l = __builtin_fma(a, w, -h);            // exact low part of the product
    const double q = __builtin_fma(h, PINV, MAGIC0) - MAGIC0;   // nearest integer to a*w/p
    const double r = __builtin_fma(-q, P, h);            // exact: |h - q p| < 2^53, integer
    return r + l;
}
// Same for |a| < 2^53 (10.285 p): q is rounded to an even integer, so
// |result| <= (1 + 0.0973 |a|/p) p.
FPF_HD double mulmod_wide(double a, double w)
{
    const double h = a * w;
    const double l = __builtin_fma(a, w, -h);
    const double q = __builtin_fma(h, PINV, MAGIC1) - MAGIC1;
    const double r = __builtin_fma(-q, P, h);
    return r + l;
}
// a*w + c mod p in one reduction (the last product of a sum reduces the whole sum: 8 operations instead of 7 + 3):
// |a| < 2^52, |w| <= p/2, c any lazy residue with |c| + (0.5 + 0.146 |a|/p) p < 2^53.  The quotient is estimated from the
// ROUNDED sum h + c (one more rounding than mulmod: 1.5 instead of 1 unit of growth), the remainder taken exactly:
// |result| <= (0.5 + 0.146 |a|/p) p.
FPF_HD double mulmod_add(double a, double w, double c)
{
    const double h = a * w;
    const double l = __builtin_fma(a, w, -h);
    const double q = __builtin_fma(h + c, PINV, MAGIC0) - MAGIC0;
    const double r = __builtin_fma(-q, P, h);            // exact: |h - q p| <= |c| + |result| + |l| < 2^53, integer
    return (r + c) + l;
}
// Same for |a| < 2^53: |result| <= (1 + 0.146 |a|/p) p
FPF_HD double mulmod_add_wide(double a, double w, double c)
{
    const double h = a * w;
    const double l = __builtin_fma(a, w, -h);
    const double q = __builtin_fma(h + c, PINV, MAGIC1) - MAGIC1;
    const double r = __builtin_fma(-q, P, h);
    return (r + c) + l;
}
// I*x mod p for any |x| < 2^53, I = zeta^2 the fourth root of unity: p = I^2 + 1, so with x = x1 I + x0 (|x0| <= I/2)
// I x = x1 I^2 + x0 I = x0 I - x1 -- four operations where a general product takes six, and the result is reduced:
// |result| <= I^2/2 + 2 I + 2^53 / I < (0.5 + 5e-7) p.  This is what makes a radix-4 butterfly (three general products
// and one by I) cheaper than four radix-2 butterflies (ntt_r4.h).
FPF_HD double mul_root4(double x)
{
    const double x1 = __builtin_fma(x, INV_ROOT4, MAGIC0) - MAGIC0;    // nearest integer to x / I (|x / I| < 2^28.2)
    const double x0 = __builtin_fma(-x1, ROOT4, x);                    // exact: small integer
    return __builtin_fma(x0, ROOT4, -x1);                              // exact: |x0 I| < 2^48.7
}
// centred residue of any |a| < 2^53: |result| <= p/2 (+1 at a tie)
FPF_HD double reduce(double a)
{
    const double q = __builtin_fma(a, PINV, MAGIC0) - MAGIC0;
    return __builtin_fma(-q, P, a);
}
// low 32 bits (two's complement) of an integer-valued double |a| < 2^51
FPF_HD uint32_t low32(double a)
{
    const double t = a + MAGIC0;                          // mantissa = 2^51 + a
    uint64_t bits;
    __builtin_memcpy(&bits, &t, 8);
    return (uint32_t)bits;
}
// torus word of a lazy residue whose true value c satisfies |c| < p/2
FPF_HD uint32_t lift_u32(double a) { return low32(reduce(a)); }
// Same for |a| < 2^51 (2.57 p) with one FP64 operation less: a = c + q p with a small
// integer q, so c mod 2^32 = low32(a) - q * (p mod 2^32).  q is read from the low mantissa
// bits of the rounding sum (two's complement), low32(a) from those of a + MAGIC0.
FPF_HD uint32_t lift_u32_small(double a)
{
    const double t = __builtin_fma(a, PINV, MAGIC0);     // mantissa = 2^51 + q
    uint64_t tb;
    __builtin_memcpy(&tb, &t, 8);
    return low32(a) - (uint32_t)tb * (uint32_t)P_U64;
}


// ---- compile-time bounds (units of p), used by static_asserts where parameters enter ----
constexpr double GROW = 0.09723;                 // p / 2^53, rounded up
constexpr double LIM_NARROW = 5.142;             // 2^52 / p, rounded down: largest |a| for mulmod
constexpr double LIM_WIDE = 10.285;              // 2^53 / p, rounded down: largest |a| for mulmod_wide / any value
constexpr double after_mulmod(double a) { return 0.5 + GROW * a; }        // |mulmod(a, w)|
constexpr double after_mulmod_wide(double a) { return 1.0 + GROW * a; }   // |mulmod_wide(a, w)|
constexpr double GROW_ADD = 0.14585;             // 1.5 p / 2^53, rounded up (mulmod_add: one more rounding in the quotient estimate)
constexpr double after_mulmod_add(double a) { return 0.5 + GROW_ADD * a; }        // |mulmod_add(a, w, c)|
constexpr double after_mulmod_add_wide(double a) { return 1.0 + GROW_ADD * a; }   // |mulmod_add_wide(a, w, c)|
constexpr double AFTER_MUL_ROOT4 = 0.5000005;    // |mul_root4(x)|

}  // namespace fpf
