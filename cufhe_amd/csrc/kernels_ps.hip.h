// kernels_ps.hip.h -- the gate path templated on a PARAMETER SET.
//
// The reference selects its TFHE parameters at compile time (CMakeLists.txt:8-24: USE_80BIT_SECURITY,
// USE_CGGI19, USE_CONCRETE -> TFHEpp params.hpp) and its kernels are templates over those structs, with
// two ring sizes (N = 1024 / N = 512, include/ntt_gpu/ntt_gpuntt.cuh:232-329) and any k
// (src/bootstrap_gpu.cu:402-421: "required for k > 1").  kernels.hip.h is hand-scheduled for the one set
// BASELINE.json names; this file is the general path: the same algorithm written once over a `PS`
// struct and instantiated for every set in kParamSets, including the default one (which lets the tests
// compare the two implementations with each other and with the oracle).
//
//   * ring: N = 1024 (ntt_wave.h, 16 coefficients per lane) or N = 512 (the stand-alone 512-point
//     transform of ntt_wave512.h, 8 per lane);
//   * k + 1 components: (k+1) l TRGSW rows, k + 1 sums;
//   * exactness: with the key read as signed 32-bit words the sum of one output coefficient is bounded by
//     (k+1) l N (Bg/2) 2^31; a set for which that exceeds p/2 (the FP64 prime of fpfield.h) takes its key
//     in `limbs` balanced limbs of `limb_bits` bits, one exact product per limb, recombined with shifts
//     mod 2^32 -- the split kernels_lvl2.hip.h uses for the 64-bit torus.  static_asserts pick nothing
//     silently: a set that does not fit does not compile.
//
// Two launch shapes, as in kernels.hip.h, both reading ONE NTT-domain key:
//   blind_rotate_ps_kernel        one workgroup of 8 waves per rotation (small launches): waves take TRGSW rows
//       round-robin (digit polynomial -> forward transform -> products with the row's (k+1) x limbs key
//       polynomials, added into LDS sums with ds_add_f64: exact integers, order-free), barrier, waves take
//       sums round-robin (inverse transform, centred lift, shifted add into the accumulator in LDS), barrier;
//   blind_rotate_ps_batch_kernel  one WAVE per rotation, 8 rotations per workgroup walking the key together
//       (large launches): the shape of blind_rotate_kernel -- accumulator and sums in registers, key rows
//       staged once per workgroup in LDS by LDS-DMA one row ahead, one barrier per row -- written over PS.
// NTT-domain key: [step][limb, HIGH limb first][row][out][q < R/2][lane][2] doubles: a TRGSW row of one limb is one contiguous
// block of (k+1) polynomials, each in the spectrum order of the wave transform (registers 2q, 2q+1 of a lane).
#pragma once
#include <type_traits>
#include "kernels.hip.h"
#include "ntt_wave512.h"

namespace cufhe_amd {

// ---- parameter sets -------------------------------------------------------------------------
struct PsDefault {      // BASELINE.json: n = 630, N = 1024, k = 1
    static constexpr const char* name = "default";
    static constexpr int n = 630, Nbit = 10, k = 1, l = 3, Bgbit = 6, t = 8, basebit = 2, limbs = 1, limb_bits = 32;
    static constexpr bool small_modulus = false;
};
struct PsK2N512 {       // k = 2 over the N = 512 ring (the shape -DUSE_CONCRETE builds in the reference)
    static constexpr const char* name = "k2n512";
    static constexpr int n = 630, Nbit = 9, k = 2, l = 3, Bgbit = 6, t = 8, basebit = 2, limbs = 1, limb_bits = 32;
    static constexpr bool small_modulus = false;
};
struct PsCggi16 {       // the original TFHE 80-bit set (-DUSE_80BIT_SECURITY): l = 2, Bg = 2^10 -> two 16-bit key limbs
    static constexpr const char* name = "cggi16";
    static constexpr int n = 500, Nbit = 10, k = 1, l = 2, Bgbit = 10, t = 8, basebit = 2, limbs = 2, limb_bits = 16;
    static constexpr bool small_modulus = false;
};
// The BASELINE numbers computed the way -DUSE_SMALL_NTT_MODULUS builds the reference (CMakeLists.txt:12,26-28;
// include/ntt_gpu/ntt_small_modulus.cuh): the external product is taken modulo P = 625 * 2^20 + 1 on a bootstrapping key whose
// torus words were switched to the P discretisation (round(a P / 2^32), src/bootstrap_gpu.cu:50-66), and every CMux increment is
// switched back (round(r 2^32 / P), include/gatebootstrapping_gpu.cuh:236-248).  Approximate by design -- the results are NOT the
// exact path's -- but deterministic integer arithmetic: the oracle compiled with -DORC_SET_SMALLMOD restates it and the words agree.
// Here the product mod P is the exact integer product reduced mod P: the switched key is centred (|w| <= P/2 < 2^29), the digits
// are the same, so the integer sums stay below (k+1) l N (Bg/2) P/2 < 2^46 and the FP64 field of fpfield.h holds them exactly --
// the transforms, key layout and launch shapes are those of `default`; only the key's preparation and the lift change.
struct PsSmallMod {
    static constexpr const char* name = "smallmod";
    static constexpr int n = 630, Nbit = 10, k = 1, l = 3, Bgbit = 6, t = 8, basebit = 2, limbs = 1, limb_bits = 32;
    static constexpr bool small_modulus = true;
};
constexpr int kParamSets = 4;

// include/ntt_gpu/ntt_small_modulus.cuh:36-68 (constants), :147-177 (the two modulus switches)
namespace smallmod {
constexpr uint32_t P = (625u << 20) + 1;                      // 655360001
constexpr uint64_t INV_MODSWITCH_MUL = (1ull << 63) / P;
// torus32_to_ntt_mod, centred: the representative of round(a P / 2^32) in (-P/2, P/2]
__host__ __device__ inline int32_t from_torus_centred(uint32_t a)
{
    const uint32_t m = (uint32_t)(((uint64_t)a * P + (1ull << 31)) >> 32);       // in [0, P]
    return m > P / 2 ? (int32_t)(m - P) : (int32_t)m;
}
// ntt_mod_to_torus32 of the residue r in [0, P)
__host__ __device__ inline uint32_t to_torus(uint32_t r) { return (uint32_t)(((uint64_t)r * INV_MODSWITCH_MUL + (1ull << 30)) >> 31); }
// a: any representative (|a| < 2^53) mod p of an exact integer sum S, |S| < p/2  ->  to_torus(S mod P)
__device__ __forceinline__ uint32_t lift(double a)
{
    const double c = fpf::reduce(a);                                   // S itself
    const double q = __builtin_rint(c * (1.0 / (double)P));
    double r = __builtin_fma(-q, (double)P, c);                        // exact: |q P| < 2^50, |r| <= P/2 + 1
    r = r < 0.0 ? r + (double)P : r;                                   // the residue in [0, P)
    return to_torus((uint32_t)(int32_t)r);
}
}  // namespace smallmod

// the limb's exact sum -> torus increment: mod 2^32 (the exact path) or switched back from the P discretisation
template <class PS>
__device__ __forceinline__ uint32_t ps_lift(double a)
{
    if constexpr (PS::small_modulus) return smallmod::lift(a);
    else return fpf::lift_u32(a);
}

template <class PS>
struct PsDims {
    static constexpr int N = 1 << PS::Nbit;
    static constexpr int R = N / 64;                        // coefficients per lane
    static constexpr int K1 = PS::k + 1;
    static constexpr int ROWS = K1 * PS::l;
    static constexpr int SUMS = K1 * PS::limbs;
    static constexpr int lvl0_words = PS::n + 1;
    static constexpr int lvl1_words = PS::k * N + 1;
    static constexpr size_t bk_step_polys = (size_t)ROWS * K1;                  // torus polynomials per CMux step
    static constexpr size_t bk_words = (size_t)PS::n * bk_step_polys * N;
    static constexpr size_t bk_ntt_step_doubles = (size_t)ROWS * SUMS * N;      // [row][out][limb][N]
    static constexpr int ks_numbase = 1 << (PS::basebit - 1);
    static constexpr size_t ksk_words = (size_t)PS::k * N * PS::t * ks_numbase * lvl0_words;
    // exactness of one limb's external product and of the unreduced LDS sums
    static constexpr double sum_bound = (double)K1 * PS::l * N * (double)(1u << (PS::Bgbit - 1)) *
                                        (PS::limbs == 1 ? 2147483648.0 : (double)(1u << (PS::limb_bits - 1)));
    static_assert(sum_bound < fpf::P / 2, "external product exceeds p/2: give the set more / narrower key limbs");
    static_assert(PS::limbs * PS::limb_bits >= 32, "key limbs do not cover the torus word");
    static_assert(ROWS * fpf::after_mulmod(0.5001) < fpf::LIM_WIDE, "row sums exceed 2^53");
    static_assert(PS::l * PS::Bgbit <= 31 && PS::t * PS::basebit <= 32 && PS::Bgbit <= 10, "decomposition out of range");
    static_assert(PS::Nbit == 9 || PS::Nbit == 10, "ring sizes: 512 and 1024");
    static_assert(!PS::small_modulus || PS::limbs == 1, "the small-modulus product is one exact product of the switched key");
    static_assert(PS::n <= 640, "abar list is sized for n <= 640");
};

template <class PS>
__host__ __device__ constexpr uint32_t ps_decomp_offset()
{
    uint32_t o = 0;
    for (int i = 1; i <= PS::l; i++) o += (1u << (PS::Bgbit - 1)) << (32 - i * PS::Bgbit);
    return o + (1u << (32 - PS::l * PS::Bgbit - 1));
}
template <class PS>
__host__ __device__ constexpr uint32_t ps_decomp_signmask()
{
    uint32_t m = 0;
    for (int i = 1; i <= PS::l; i++) m |= (1u << (PS::Bgbit - 1)) << (32 - i * PS::Bgbit);
    return m;
}

// ---- one polynomial per wave, either ring size ------------------------------------------------
template <int NBIT> struct Poly;
template <> struct Poly<10> {
    static constexpr int R = 16, tile_bytes = kTileBytes, table_bytes = kLdsTableBytes;
    using Tables = NttTables;
    using Ctx = WaveCtx;
    static __device__ __forceinline__ void load_tables(char* lds, const Tables* gt) { load_tables_to_lds((double*)lds, gt); }
    static __device__ __forceinline__ Ctx ctx(char* lds, int tile_off, int tables_off, const Tables* gt, int lane)
    {
        return make_wave_ctx(lds, tile_off, tables_off, gt, lane);
    }
    // natural order in (element lane + 64 r in register r), spectrum order out; any |x| < 2^32
    static __device__ __forceinline__ void forward(double (&x)[R], const Ctx& c) { ntt_forward<false>(x, c); }
    // gadget digits (|x| <= Bg/2, as long as the exact radix-4 butterfly of the first two stages stays exact: Bg <= 2^10 does): spectrum
    // bound 6.18 p (Bg = 2^6) .. 6.3 p instead of 8.92 p
    static __device__ __forceinline__ void forward_small(double (&x)[R], const Ctx& c) { ntt_forward<true>(x, c); }
    static constexpr bool small_ok(double digit_max) { return forward_digit_spectrum_bound(digit_max) > 0; }
    static constexpr double spectrum_bound(bool small, double digit_max) { return small ? forward_digit_spectrum_bound(digit_max) : forward_words_spectrum_bound(); }
    static __device__ __forceinline__ void inverse(double (&x)[R], const Ctx& c) { ntt_inverse(x, c); }
};
template <> struct Poly<9> {
    static constexpr int R = 8, tile_bytes = kTile512Bytes, table_bytes = kLds512TableBytes;
    using Tables = Ntt512Tables;        // the stand-alone 512-point negacyclic transform (capi.hip: tables512[2])
    using Ctx = Wave512Ctx;
    static __device__ __forceinline__ void load_tables(char* lds, const Tables* gt)
    {
        const double* src = gt->tb_fwd;          // tb_fwd .. tc_inv are contiguous
        for (int i = threadIdx.x; i < kLds512TableDoubles; i += blockDim.x) ((double*)lds)[i] = src[i];
    }
    static __device__ __forceinline__ Ctx ctx(char* lds, int tile_off, int tables_off, const Tables* gt, int lane)
    {
        return make_wave512_ctx(lds, tile_off, tables_off, gt, lane);
    }
    static __device__ __forceinline__ void forward(double (&x)[R], const Ctx& c) { ntt512_forward(x, c); }
    // |x| <= 32: stages 0 and 1 exact (the stand-alone transform's first roots are I, zeta, zeta^3)
    static __device__ __forceinline__ void forward_small(double (&x)[R], const Ctx& c) { ntt512_forward_small(x, c); }
    static constexpr bool small_ok(double digit_max) { return digit_max <= 32.0; }
    static constexpr double spectrum_bound(bool, double) { return 7.23; }      // ntt_wave512.h: inputs far below p, |out| <= 7.22 p
    static __device__ __forceinline__ void inverse(double (&x)[R], const Ctx& c) { ntt512_inverse(x, c); }
};

// The transforms of both blind-rotate kernels: for the 1024-point ring the radix-4 passes of ntt_r4.h on the r4 tables in their
// packed form (the key conversion keeps the radix-2 transform and tables above; the spectrum order is the same), for the 512-point
// ring Poly<9> as it is.  DIGIT_MAX = Bg/2.
template <int NBIT, int DIGIT_MAX> struct PsbPoly;
template <int DIGIT_MAX> struct PsbPoly<9, DIGIT_MAX> : Poly<9> {
    static constexpr bool kR4Tables = false;
    static constexpr bool kSmall = Poly<9>::small_ok((double)DIGIT_MAX);
    static constexpr double kSpectrum = Poly<9>::spectrum_bound(kSmall, (double)DIGIT_MAX);
    struct State {};                 // per-kernel transform state (nothing for this ring)
    static __device__ __forceinline__ void init(State&, const Tables*) {}
    static __device__ __forceinline__ void forward_digits(double (&x)[R], const Ctx& c, const State&)
    {
        if (kSmall) Poly<9>::forward_small(x, c);
        else Poly<9>::forward(x, c);
    }
};
template <class PS>
using PsbPolyOf = PsbPoly<PS::Nbit, (1 << (PS::Bgbit - 1))>;
template <int DIGIT_MAX> struct PsbPoly<10, DIGIT_MAX> {
    static constexpr bool kR4Tables = true;
    static constexpr int R = 16, tile_bytes = kTileBytes, table_bytes = kLdsTablePackedBytes;
    using Tables = NttTables;         // the r4 instance (capi.hip: DeviceState::tables_r4)
    using Ctx = WaveCtx;
    using Fwd = r4::FwdDigits<DIGIT_MAX>;
    using Inv = r4::Inverse<r4::Uniform<501>>;
    static_assert(r4::valid(Fwd::Spectrum::in()), "radix-4 forward transform of this set's gadget digits: a value exceeds 2^53");
    static constexpr double kSpectrum = r4::max_of(Fwd::Spectrum::in());
    static __device__ __forceinline__ void load_tables(char* lds, const Tables* gt) { load_packed_tables_to_lds((double*)lds, gt); }
    static __device__ __forceinline__ Ctx ctx(char* lds, int tile_off, int tables_off, const Tables* gt, int lane)
    {
        return make_wave_ctx_packed(lds, tile_off, tables_off, gt, lane);
    }
    struct State {};
    static __device__ __forceinline__ void init(State&, const Tables*) {}
    static __device__ __forceinline__ void forward_digits(double (&x)[R], const Ctx& c, const State&)
    {
        // (blind_rotate_kernel's store-inside-the-pass form with pinned scalar twiddles measured slower here: 38.4 against 37.5 ms
        // for `default`, 45.6 against 45.0 for `cggi16` -- 105 SGPRs)
        ntt_forward_digits_a_r4<DIGIT_MAX>(x, c);
        ntt_forward_digits_bc_r4<DIGIT_MAX>(x, c);
    }
    // x reduced (|x| <= p/2) in, any |out| < 2^53
    static __device__ __forceinline__ void inverse(double (&x)[R], const Ctx& c) { ntt_inverse_r4<r4::Uniform<501>>(x, c); }
};

// The transforms of the workgroup-per-rotation kernel: those of the wave-per-rotation kernel (radix-4 passes for the 1024-point ring;
// one rotation 4.97 -> 4.47 ms on `default`, 3.93 -> 3.70 on `cggi16`, 3.51 -> 3.36 on `k2n512` through the digit form of its
// 512-point transform: profiles/r05_ps_wg_radix4_ab.txt); -DCUFHE_AMD_PS_WG_RADIX2 (tools/build_variant.py) keeps the radix-2
// transform the key conversion uses, for A/B runs.
#ifdef CUFHE_AMD_PS_WG_RADIX2
constexpr bool kPsWgR4 = false;
#else
constexpr bool kPsWgR4 = true;
#endif
template <class PS> using PsWgPolyOf = std::conditional_t<kPsWgR4, PsbPolyOf<PS>, Poly<PS::Nbit>>;

// waves of the workgroup-per-rotation kernel: one per TRGSW row where that takes more than 8 (k2n512: 9 rows would
// otherwise be walked in two passes, the second with one busy wave)
template <class PS> constexpr int kPsWavesOf = ((PS::k + 1) * PS::l > 8) ? 12 : 8;

template <class PS>
struct PsLds {
    using D = PsDims<PS>;
    using PO = PsWgPolyOf<PS>;
    static constexpr int tables = 0;
    static constexpr int tiles = tables + PO::table_bytes;
    static constexpr int waves = kPsWavesOf<PS>;
    static constexpr int threads = 64 * waves;
    static constexpr int acc = tiles + waves * PO::tile_bytes;                 // [K1][N] u32
    static constexpr int sums = (acc + D::K1 * D::N * 4 + 15) & ~15;           // [SUMS][R][64] f64
    static constexpr int abar = sums + D::SUMS * D::N * 8;
    static constexpr int bytes = abar + kAbarBytes + 16;
    static_assert(bytes <= 160 * 1024, "parameter set does not fit the CU's LDS");
};

// ----------------------------------------------------------------------------------------------
// BK (torus words) -> NTT domain, one wave per (polynomial, limb).
// bk: [step][row][out][N] u32 (TFHEpp's BootstrappingKey layout, src/bootstrap_gpu.cu:43-49);
// bk_ntt: [step][limb][row][out][R/2][64][2] doubles, scaled by N^-1, centred.
// ----------------------------------------------------------------------------------------------
template <class PS>
__global__ __launch_bounds__(kNttThreads) void bk_to_ntt_ps_kernel(
    double* __restrict__ bk_ntt, const uint32_t* __restrict__ bk, size_t polys,
    const typename Poly<PS::Nbit>::Tables* __restrict__ gt, double n_inverse)
{
    using D = PsDims<PS>;
    using PO = Poly<PS::Nbit>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PO::load_tables(smem, gt);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * kNttWavesPerBlock + wave;
    if (w >= polys * PS::limbs) return;
    const size_t poly = w / PS::limbs;
    const int limb = (int)(w % PS::limbs);
    const typename PO::Ctx ctx = PO::ctx(smem, PO::table_bytes + wave * PO::tile_bytes, 0, gt, lane);
    double x[PO::R];
#pragma unroll
    for (int r = 0; r < PO::R; r++) {
        int64_t s = PS::small_modulus ? smallmod::from_torus_centred(bk[poly * D::N + lane + 64 * r]) : (int32_t)bk[poly * D::N + lane + 64 * r];
        if (PS::limbs > 1) {      // balanced limbs: s = sum_i limb_i 2^(i limb_bits), the top limb takes the rest
            int64_t v = 0;
            for (int m = 0; m <= limb; m++) {
                v = (m == PS::limbs - 1) ? s : (int64_t)((uint64_t)s << (64 - PS::limb_bits)) >> (64 - PS::limb_bits);
                s = (s - v) >> PS::limb_bits;
            }
            x[r] = (double)v;
        } else {
            x[r] = (double)s;
        }
    }
    PO::forward(x, ctx);
    const size_t step = poly / D::bk_step_polys, in_step = poly % D::bk_step_polys;       // in_step = row * K1 + out
    // the limbs of a step are stored HIGH limb first: the wave-per-rotation kernel walks them in storage order and then only has to
    // keep the top 32 - limb_bits bits of what the earlier limbs contributed (blind_rotate_ps_batch_kernel, `delta`)
    double2* dst = (double2*)(bk_ntt + ((step * PS::limbs + (PS::limbs - 1 - limb)) * D::bk_step_polys + in_step) * D::N);
#pragma unroll
    for (int q = 0; q < PO::R / 2; q++) {
        double2 v;
        v.x = fpf::reduce(fpf::mulmod_wide(x[2 * q], n_inverse));
        v.y = fpf::reduce(fpf::mulmod_wide(x[2 * q + 1], n_inverse));
        dst[q * 64 + lane] = v;
    }
}

// ----------------------------------------------------------------------------------------------
// Blind rotate (+ sample extract at 0), one workgroup per rotation.
// descs: in0/in1 lvl0 TLWEs (n + 1 words), out a lvl1 TLWE (k N + 1 words) or null; acc_dump (optional)
// receives the raw accumulator ((k+1) N words).  include/gatebootstrapping_gpu.cuh:29-52,115-345,
// src/bootstrap_gpu.cu:366-381.
// ----------------------------------------------------------------------------------------------
template <class PS>
__global__ __launch_bounds__(64 * kPsWavesOf<PS>) void blind_rotate_ps_kernel(
    const LinDesc* __restrict__ descs, int count, const double* __restrict__ bk_ntt,
    const typename Poly<PS::Nbit>::Tables* __restrict__ gt, int steps, uint32_t* __restrict__ acc_dump)
{
    using D = PsDims<PS>;
    using PO = PsWgPolyOf<PS>;
    using L = PsLds<PS>;
    constexpr int N = D::N, R = D::R, K1 = D::K1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int g = blockIdx.x;
    if (g >= count) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    PO::load_tables(smem + L::tables, gt);
    uint32_t* accL = (uint32_t*)(smem + L::acc);
    double* sumL = (double*)(smem + L::sums);
    uint16_t* abar_lds = (uint16_t*)(smem + L::abar);
    uint32_t* bbar_slot = (uint32_t*)(smem + L::abar + kAbarBytes);

    const LinDesc d = descs[g];
    for (int i = tid; i <= PS::n; i += L::threads) {      // pre-add and modulus switch, :316-345
        const uint32_t c = (uint32_t)d.ca * d.in0[i] + (uint32_t)d.cb * d.in1[i];
        if (i < PS::n) abar_lds[i] = (uint16_t)((c + (1u << (32 - 2 - PS::Nbit))) >> (32 - 1 - PS::Nbit));
        else *bbar_slot = 2 * N - ((c + d.off) >> (32 - 1 - PS::Nbit));
    }
    for (int i = tid; i < D::SUMS * N; i += L::threads) sumL[i] = 0.0;
    __syncthreads();
    {   // RotatedTestVector, :29-52: mask components zero, body +-mu
        const uint32_t bbar = *bbar_slot;
        for (int e = tid; e < N; e += L::threads) {
            const bool neg = (bbar != 2 * N) && (((uint32_t)e < (bbar & (N - 1))) != ((bbar >> PS::Nbit) != 0));
            for (int j = 0; j < PS::k; j++) accL[j * N + e] = 0;
            accL[PS::k * N + e] = neg ? 0u - kMu : kMu;
        }
    }
    __syncthreads();
    const typename PO::Ctx ctx = PO::ctx(smem, L::tiles + wave * PO::tile_bytes, L::tables, gt, lane);

    // One TRGSW row per wave (waves >= rows).  The key polynomials of the wave's row -- (k+1) x limbs of them, R/2 double2 per lane
    // each -- are read a whole step ahead: issued after the products of step i, they arrive while the inverse transforms of step i
    // run on other waves, so the L2 latency of the key stream is off the step's critical path: one rotation 4.47 -> 4.16 ms on
    // `default`, 3.36 -> 3.03 on `k2n512`, 3.70 -> 3.46 on `cggi16` (profiles/r05_ps_wg_radix4_ab.txt; with every CU busy the
    // gain is 0 - 3 %).  -DCUFHE_AMD_PS_WG_NO_PREFETCH: loads where they are used, for A/B runs.
    static_assert(L::waves >= D::ROWS, "one TRGSW row per wave");
    const bool has_row = wave < D::ROWS;
    const int row = wave, j = row / PS::l, dg = row % PS::l;
    double2 kb[D::SUMS][R / 2];
    auto load_key = [&](int step) {
        if (!has_row || step >= steps) return;
#pragma unroll
        for (int s = 0; s < D::SUMS; s++) {
            const int out = s / PS::limbs, limb = s % PS::limbs;
            const double2* kp = (const double2*)(bk_ntt + ((((size_t)step * PS::limbs + (PS::limbs - 1 - limb)) * D::ROWS + row) * K1 + out) * N) + lane;
#pragma unroll
            for (int q = 0; q < R / 2; q++) kb[s][q] = kp[q * 64];
        }
    };
#ifndef CUFHE_AMD_PS_WG_NO_PREFETCH
    load_key(0);
#endif
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_PHASES)
    // timing-only (tools/ps_phases.py): cycles this wave spends in each phase of a step
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#define CUFHE_AMD_PS_PHASE(k) { const unsigned long long tn = __builtin_readcyclecounter(); ph[k] += tn - tc; tc = tn; }
#else
#define CUFHE_AMD_PS_PHASE(k)
#endif

#pragma unroll 1
    for (int i = 0; i < steps; i++) {
        const uint32_t abar = __builtin_amdgcn_readfirstlane((uint32_t)abar_lds[i]);
        const int alo = (int)(abar & (N - 1));
        const bool ahi = (abar >> PS::Nbit) != 0;
        if (has_row) {
            const uint32_t* accj = accL + j * N;
            const uint32_t pos = 32 - (dg + 1) * PS::Bgbit;
            double x[R];
#pragma unroll
            for (int r = 0; r < R; r++) {        // (X^abar - 1) acc_j, decomposed: :157-181
                const int e = lane + 64 * r;
                const uint32_t rot = accj[(e - alo) & (N - 1)];
                const bool neg = (e < alo) != ahi;
                const uint32_t t = ((neg ? 0u - rot : rot) - accj[e] + ps_decomp_offset<PS>()) ^ ps_decomp_signmask<PS>();
                x[r] = (double)(int32_t)__builtin_amdgcn_sbfe(t, pos, (uint32_t)PS::Bgbit);
            }
            CUFHE_AMD_PS_PHASE(0)
            if constexpr (kPsWgR4) PO::forward_digits(x, ctx, typename PO::State{});
            else PO::forward(x, ctx);
#pragma unroll
            for (int r = 0; r < R; r++) x[r] = fpf::reduce(x[r]);
            CUFHE_AMD_PS_PHASE(1)
#ifdef CUFHE_AMD_PS_WG_NO_PREFETCH
            load_key(i);
#endif
#pragma unroll
            for (int s = 0; s < D::SUMS; s++) {      // :206-221, one product per (output component, key limb)
                double* sp = sumL + s * N + lane;
#pragma unroll
                for (int q = 0; q < R / 2; q++) {
                    __hip_atomic_fetch_add(sp + (2 * q) * 64, fpf::mulmod(x[2 * q], kb[s][q].x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(sp + (2 * q + 1) * 64, fpf::mulmod(x[2 * q + 1], kb[s][q].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            CUFHE_AMD_PS_PHASE(2)
        }
#ifndef CUFHE_AMD_PS_WG_NO_PREFETCH
        load_key(i + 1);
#endif
        __syncthreads();
        CUFHE_AMD_PS_PHASE(3)
#pragma unroll 1
        for (int s = wave; s < D::SUMS; s += L::waves) {      // :227-284
            double* sp = sumL + s * N + lane;
            double A[R];
#pragma unroll
            for (int r = 0; r < R; r++) { A[r] = fpf::reduce(sp[r * 64]); sp[r * 64] = 0.0; }
            CUFHE_AMD_PS_PHASE(4)
            PO::inverse(A, ctx);
            CUFHE_AMD_PS_PHASE(5)
            const int out = s / PS::limbs, shl = (s % PS::limbs) * PS::limb_bits;
            uint32_t* acck = accL + out * N + lane;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t v = ps_lift<PS>(A[r]) << shl;         // the limb's exact sum, shifted, mod 2^32
                if (PS::limbs == 1) acck[64 * r] += v;
                else __hip_atomic_fetch_add(acck + 64 * r, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            CUFHE_AMD_PS_PHASE(6)
        }
        __syncthreads();
        CUFHE_AMD_PS_PHASE(7)
    }

    if (acc_dump) {
        uint32_t* o = acc_dump + (size_t)g * K1 * N;
        for (int e = tid; e < K1 * N; e += L::threads) o[e] = accL[e];
    }
#if defined(CUFHE_AMD_DIAGNOSTIC_BUILD) && defined(CUFHE_AMD_ABL_PHASES)
    if (acc_dump && lane == 0 && g == 0) {
        __syncthreads();
        unsigned long long* o = (unsigned long long*)acc_dump + 16 + wave * 8;      // overwrites part of the dump: timing only
        for (int k = 0; k < 8; k++) o[k] = ph[k];
    }
#endif
    if (d.out) {   // __SampleExtractIndex__<P,0>: per mask component a'[0] = a[0], a'[m] = -a[N-m]; b' = b[0]
        uint32_t* o = d.out;
        for (int e = tid; e < PS::k * N; e += L::threads) {
            const int j = e / N, m = e % N;
            o[e] = m == 0 ? accL[j * N] : 0u - accL[j * N + N - m];
        }
        if (tid == 0) o[PS::k * N] = accL[PS::k * N];
    }
}

// ----------------------------------------------------------------------------------------------
// Blind rotate (+ sample extract at 0), one WAVE per rotation, 8 rotations per workgroup: the design of
// blind_rotate_kernel (kernels.hip.h) over PS.  Per wave: the accumulator ((k+1) x R words) and the k+1 sums
// ((k+1) x R doubles) live in registers; X^abar acc_j goes through the wave's tile (rotate_sub); a TRGSW row of
// the current key limb ((k+1) polynomials, 16 or 12 KiB) is copied once per workgroup into one of three LDS
// buffers by LDS-DMA, one row ahead, and every wave takes exactly one barrier per row -- waves 4-7 before the
// forward transform of the row, waves 0-3 after it, so that the two waves of a SIMD are half a row apart
// (kernels.hip.h, RowPipe).  A set with key limbs walks the rows of a step once per limb: the digits are
// decomposed again from the unchanged accumulator and the limb's exact sums, lifted and shifted, are collected
// in `delta` until the last limb is done.
// ----------------------------------------------------------------------------------------------
// rotations (waves) per workgroup: 8 = two per SIMD.  The N = 512 sets need only 145 VGPRs and could run 12 (three per
// SIMD): measured 107.8 k against 102 k gates/s at multiples of 3072 rotations, but a launch of 4096 then pays a second
// round a third full (76.9 k), so every set keeps rounds of 2048.
template <class PS> constexpr int kPsbWavesOf = 8;
constexpr int kPsbRowBuffers = 3;

template <class PS>
struct PsbLds {
    using D = PsDims<PS>;
    using PO = PsbPolyOf<PS>;
    static constexpr int waves = kPsbWavesOf<PS>;
    static constexpr int threads = 64 * waves;
    static constexpr int row_bytes = D::K1 * D::N * 8;
    static constexpr int row_pieces = row_bytes / 1024;                        // LDS-DMA pieces of 1 KiB
    static constexpr int rows = 0;                                             // row buffers first: DS offsets < 64 KiB
    static constexpr int tables = rows + kPsbRowBuffers * row_bytes;
    static constexpr int tiles = tables + PO::table_bytes;
    static constexpr int abar = tiles + waves * PO::tile_bytes;
    static constexpr int bytes = abar + waves * kAbarBytes;
    static_assert(row_bytes % 1024 == 0, "row is moved in 1 KiB pieces");
    static_assert(2 * D::N * 4 <= PO::tile_bytes, "rotate_sub needs 2N words of the wave's tile");
    static_assert(bytes <= 160 * 1024, "parameter set does not fit the CU's LDS");
};

template <int ROW_BYTES, int WAVES>
struct PsRowPipe {
    const char* bk;
    char* buf;
    int wave, lane, total_rows;
    bool late;
    __device__ __forceinline__ void issue(int R) const
    {
        if (R >= total_rows) return;
        const char* src = bk + (size_t)R * ROW_BYTES + lane * 16;
        char* dst = buf + (R % kPsbRowBuffers) * ROW_BYTES;
        constexpr int pieces = ROW_BYTES / 1024;
#pragma unroll
        for (int c = 0; c < (pieces + WAVES - 1) / WAVES; c++) {
            const int piece = wave + WAVES * c;
            if (piece < pieces)
                lds_dma16(src + piece * 1024, dst + piece * 1024);
        }
    }
    __device__ __forceinline__ void sync(int R) const
    {
        lds_dma_wait_all();  // this wave's pieces of row R (ntt_wave.h: lds_dma16, invisible to the compiler's own counting)
        __syncthreads();     // every wave's pieces of row R have landed
        issue(R + 1);        // into the buffer of row R - 2, whose last reader passed its barrier R - 1
    }
    __device__ __forceinline__ const char* row(int R) const
    {
        return buf + opaque((R % kPsbRowBuffers) * ROW_BYTES + lane * 16);
    }
};

// acc_j -> (X^abar - 1) acc_j + gadget offset, through the wave's tile (kernels.hip.h, rotate_sub)
template <class PS, int R>
__device__ __forceinline__ void ps_rotate_sub(uint32_t (&temp)[R], const uint32_t (&acc)[R], char* tile, int lane, uint32_t abar)
{
    constexpr int N = 1 << PS::Nbit;
    const int alo = (int)(abar & (N - 1));
    const bool ahi = (abar >> PS::Nbit) != 0;
    char* wpos = tile + opaque(4 * lane + (ahi ? 0 : 4 * N));
    char* wneg = tile + opaque(4 * lane + (ahi ? 4 * N : 0));
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; r++) {
        *(uint32_t*)(wpos + 256 * r) = acc[r];
        *(uint32_t*)(wneg + 256 * r) = 0u - acc[r];
    }
    asm volatile("" ::: "memory");
    const char* rbase = tile + opaque(4 * (lane - alo + N));
    uint32_t rot[R];
#pragma unroll
    for (int r = 0; r < R; r++) rot[r] = *(const uint32_t*)(rbase + 256 * r);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; r++) temp[r] = (rot[r] - acc[r] + ps_decomp_offset<PS>()) ^ ps_decomp_signmask<PS>();
}

template <class PS>
__global__ __launch_bounds__(64 * kPsbWavesOf<PS>, kPsbWavesOf<PS> / 4) void blind_rotate_ps_batch_kernel(
    const LinDesc* __restrict__ descs, int count, const double* __restrict__ bk_ntt,
    const typename Poly<PS::Nbit>::Tables* __restrict__ gt, int steps, uint32_t* __restrict__ acc_dump)
{
    using D = PsDims<PS>;
    using PO = PsbPolyOf<PS>;
    using L = PsbLds<PS>;
    constexpr int N = D::N, R = D::R, K1 = D::K1;
    constexpr double kSpec = PO::kSpectrum;
    constexpr double kRowTerm = fpf::after_mulmod_wide(kSpec);
    constexpr bool kAllRowsFit = D::ROWS * kRowTerm < fpf::LIM_WIDE;
    static_assert(kSpec > 0 && kSpec < fpf::LIM_WIDE, "digit spectrum exceeds what the wide product accepts");
    static_assert(0.5001 + PS::l * kRowTerm < fpf::LIM_WIDE, "the rows of one component do not fit on top of a reduced sum");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PO::load_tables(smem + L::tables, gt);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * L::waves + wave;
    char* tile = smem + L::tiles + wave * PO::tile_bytes;
    uint16_t* abar_lds = (uint16_t*)(smem + L::abar + wave * kAbarBytes);
    const typename PO::Ctx ctx = PO::ctx(smem, L::tiles + wave * PO::tile_bytes, L::tables, gt, lane);
    const PsRowPipe<L::row_bytes, L::waves> pipe{(const char*)bk_ntt, smem + L::rows, wave, lane, steps * PS::limbs * D::ROWS, wave >= L::waves / 2};
    pipe.issue(0);
    if (g >= count) {
        // no rotation for this wave (tail of the batch): it only keeps its share of the row pipeline going
        __syncthreads();
#pragma unroll 1
        for (int Rr = 0; Rr < pipe.total_rows; Rr++) pipe.sync(Rr);
        return;
    }

    const LinDesc d = descs[g];
    uint32_t bword = 0;       // pre-add and modulus switch, :316-345
    for (int i = lane; i <= PS::n; i += 64) {
        const uint32_t c = (uint32_t)d.ca * d.in0[i] + (uint32_t)d.cb * d.in1[i];
        if (i < PS::n) abar_lds[i] = (uint16_t)((c + (1u << (32 - 2 - PS::Nbit))) >> (32 - 1 - PS::Nbit));
        else bword = c;
    }
    bword = __builtin_amdgcn_readlane(bword, PS::n % 64) + d.off;
    const uint32_t bbar = 2 * N - (bword >> (32 - 1 - PS::Nbit));
    uint32_t acc[K1][R];      // RotatedTestVector, :29-52: mask components zero, body +-mu
#pragma unroll
    for (int r = 0; r < R; r++) {
        const uint32_t e = lane + 64 * r;
        const bool neg = (bbar != 2 * N) && ((e < (bbar & (N - 1))) != ((bbar >> PS::Nbit) != 0));
#pragma unroll
        for (int j = 0; j < PS::k; j++) acc[j][r] = 0;
        acc[PS::k][r] = neg ? 0u - kMu : kMu;
    }
    __syncthreads();          // tables staged; abar list visible
    typename PO::State po_state;
    PO::init(po_state, gt);

    int row_id = 0;
#pragma unroll 1
    for (int i = 0; i < steps; i++) {
        const uint32_t abar = __builtin_amdgcn_readfirstlane((uint32_t)abar_lds[i]);
        // Sets with key limbs: the limbs are walked from the highest down (the key is stored that way), so what the limbs before the last
        // contribute is a multiple of 2^limb_bits -- with limb_bits >= 16 only its top 16 bits have to be carried through the last
        // limb's row walk, two coefficients (r, r + R/2) to a register: R/2 registers per component where there were R (the full words
        // were what spilled: 19 registers of scratch on the 80-bit set).  The last limb (shift 0) goes straight into the accumulator.
#ifdef CUFHE_AMD_PS_NO_PACK      // experiment (tools/build_variant.py): the full words carried (profiles/r05_lvl2_ab.txt)
        constexpr bool kPackDelta = false;
#else
        constexpr bool kPackDelta = PS::limbs > 1 && PS::limb_bits >= 16;
#endif
        constexpr int kDeltaRegs = PS::limbs == 1 ? 1 : kPackDelta ? R / 2 : R;
        uint32_t delta[K1][kDeltaRegs];      // only live for sets with key limbs
        if (PS::limbs > 1) {
#pragma unroll
            for (int o = 0; o < K1; o++)
#pragma unroll
                for (int r = 0; r < kDeltaRegs; r++) delta[o][r] = 0;
        }
#pragma unroll 1
        for (int slot = 0; slot < PS::limbs; slot++) {
            const int limb = PS::limbs - 1 - slot;
            double A[K1][R];
#pragma unroll
            for (int o = 0; o < K1; o++)
#pragma unroll
                for (int r = 0; r < R; r++) A[o][r] = 0.0;
#pragma unroll
            for (int j = 0; j < K1; j++) {       // include/gatebootstrapping_gpu.cuh:153-224
                uint32_t temp[R];
                ps_rotate_sub<PS, R>(temp, acc[j], tile, lane, abar);
#pragma unroll 1
                for (int dg = 0; dg < PS::l; dg++, row_id++) {
                    const uint32_t pos = 32 - (dg + 1) * PS::Bgbit;
                    double x[R];
#pragma unroll
                    for (int r = 0; r < R; r++) x[r] = (double)(int32_t)__builtin_amdgcn_sbfe(temp[r], pos, (uint32_t)PS::Bgbit);
                    if (pipe.late) pipe.sync(row_id);
                    PO::forward_digits(x, ctx, po_state);
                    if (!pipe.late) pipe.sync(row_id);
                    // :206-221.  The spectrum (<= kSpec p) goes into wide products unreduced: each is <= kRowTerm p, the l
                    // rows of a component add up below 2^53 on top of a reduced sum (static_assert above), and a set whose
                    // (k+1) l rows do not fit altogether reduces the sums between components (below).  Row pieces are
                    // read kDepth ahead of their use; the sched_barrier keeps the DS reads where they are written.
                    const char* rowp = pipe.row(row_id);
                    constexpr int kPieces = K1 * (R / 2), kDepth = 3;
                    double2 b[kPieces];
#pragma unroll
                    for (int q = 0; q < kDepth; q++) b[q] = *(const double2*)(rowp + q * 1024);
#pragma unroll
                    for (int q = 0; q < kPieces; q++) {
                        if (q + kDepth < kPieces) b[q + kDepth] = *(const double2*)(rowp + (q + kDepth) * 1024);
                        __builtin_amdgcn_sched_barrier(0x0006);
                        const int o = q / (R / 2), qq = q % (R / 2);
                        A[o][2 * qq] += fpf::mulmod_wide(x[2 * qq], b[q].x);
                        A[o][2 * qq + 1] += fpf::mulmod_wide(x[2 * qq + 1], b[q].y);
                        __builtin_amdgcn_sched_barrier(0x0006);
                    }
                }
                if (!kAllRowsFit && j + 1 < K1) {
#pragma unroll
                    for (int o = 0; o < K1; o++)
#pragma unroll
                        for (int r = 0; r < R; r++) A[o][r] = fpf::reduce(A[o][r]);
                }
            }
            const int shl = limb * PS::limb_bits;
#pragma unroll
            for (int o = 0; o < K1; o++) {               // :227-284
#pragma unroll
                for (int r = 0; r < R; r++) A[o][r] = fpf::reduce(A[o][r]);
                PO::inverse(A[o], ctx);
                uint32_t v[R];
#pragma unroll
                for (int r = 0; r < R; r++) v[r] = ps_lift<PS>(A[o][r]);         // the limb's exact sum mod 2^32
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (PS::limbs == 1) {
                        acc[o][r] += v[r];
                    } else if (limb > 0) {                         // v << shl is a multiple of 2^limb_bits
                        if (kPackDelta && PS::limbs == 2) {
                            // the only limb before the last: bits 16..31 of (v << 16) are v's low half -- one and, one shift-or per pair
                            if (r < R / 2) delta[o][r] = (v[r] & 0xFFFFu) | (v[r + R / 2] << 16);
                        } else if (kPackDelta) {
                            const uint32_t top = (v[r] << shl) >> 16;              // bits 16..31 of the contribution
                            if (r < R / 2) delta[o][r] = (delta[o][r] & 0xFFFF0000u) | ((delta[o][r] + top) & 0xFFFFu);
                            else delta[o][r - R / 2] += top << 16;
                        } else {
                            delta[o][r] += v[r] << shl;
                        }
                    } else {                                       // the last limb: everything into the accumulator
                        const uint32_t carried = !kPackDelta ? delta[o][r] : r < R / 2 ? delta[o][r] << 16 : delta[o][r - R / 2] & 0xFFFF0000u;
                        acc[o][r] += v[r] + carried;
                    }
                }
            }
        }
    }

    if (acc_dump) {
        uint32_t* o = acc_dump + (size_t)g * K1 * N;
#pragma unroll
        for (int j = 0; j < K1; j++)
#pragma unroll
            for (int r = 0; r < R; r++) o[j * N + lane + 64 * r] = acc[j][r];
    }
    if (d.out) {   // __SampleExtractIndex__<P,0>: per mask component a'[0] = a[0], a'[m] = -a[N-m]; b' = b[0]
        uint32_t* o = d.out;
#pragma unroll
        for (int j = 0; j < PS::k; j++)
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = lane + 64 * r;
                if (e == 0) o[j * N] = acc[j][r];
                else o[j * N + N - e] = 0u - acc[j][r];
            }
        if (lane == 0) o[PS::k * N] = acc[PS::k][0];
    }
}

// __SampleExtractIndex__<P, 0> (src/bootstrap_gpu.cu:366-381) with one descriptor per TRLWE: in0 = the TRLWE ((k+1) N words),
// out = a lvl1 TLWE (k N + 1 words) -- the first step of __SEIandKS__ / __SEIandBootstrap2TRLWE__ on a set
template <class PS>
__global__ __launch_bounds__(256) void sample_extract_ps_kernel(const LinDesc* __restrict__ descs, int count)
{
    constexpr int N = 1 << PS::Nbit, KN = PS::k * N;
    for (int g = blockIdx.x; g < count; g += gridDim.x) {
        const uint32_t* in = descs[g].in0;
        uint32_t* o = descs[g].out;
        for (int e = threadIdx.x; e <= KN; e += blockDim.x) {
            const int j = e / N, m = e % N;
            o[e] = e == KN ? in[KN] : (m == 0 ? in[j * N] : 0u - in[j * N + N - m]);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// CMUXNTT on a set (__CMUXNTT__, src/bootstrap_gpu.cu:197-285; TRLWESubAndDecomposition :162-195): res = c0 + trgsw [x] (c1 - c0),
// one wave per CMUX, operands through descriptors (what the stream scheduler launches for the CMUXNTT calls of one dependence
// level).  trgsw_ntt is one "step" of bk_to_ntt_ps_kernel's layout: [limb, HIGH first][row][out][R/2][64][2] doubles -- what
// TRGSW2NTT on the set writes (the reference's template runs on whatever lvl1param the build selected, :75-94).  A set with key
// limbs walks the rows once per limb; each limb's exact sum is lifted, shifted and added mod 2^32.  res may be c0 or c1: both are
// read in full before the first word of res is written.
// ----------------------------------------------------------------------------------------------
template <class PS>
__global__ __launch_bounds__(kNttThreads) void cmux_desc_ps_kernel(const CmuxDesc* __restrict__ descs, int count,
                                                                   const typename Poly<PS::Nbit>::Tables* __restrict__ gt)
{
    using D = PsDims<PS>;
    using PO = Poly<PS::Nbit>;
    constexpr int N = D::N, R = D::R, K1 = D::K1;
    static_assert(!PS::small_modulus, "the reference's small-modulus build has no CMUXNTT (src/cufhe_gates_gpu.cu:68-86)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PO::load_tables(smem, gt);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * kNttWavesPerBlock + wave;
    if (g >= count) return;
    const typename PO::Ctx ctx = PO::ctx(smem, PO::table_bytes + wave * PO::tile_bytes, 0, gt, lane);
    const CmuxDesc d = descs[g];
    uint32_t temp[K1][R], resw[K1][R];
#pragma unroll
    for (int j = 0; j < K1; j++)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int e = j * N + lane + 64 * r;
            const uint32_t a0 = d.c0[e];
            resw[j][r] = a0;
            temp[j][r] = (d.c1[e] - a0 + ps_decomp_offset<PS>()) ^ ps_decomp_signmask<PS>();
        }
#pragma unroll 1
    for (int slot = 0; slot < PS::limbs; slot++) {          // storage order: the highest limb first
        const int limb = PS::limbs - 1 - slot;
        double A[K1][R];
#pragma unroll
        for (int o = 0; o < K1; o++)
#pragma unroll
            for (int r = 0; r < R; r++) A[o][r] = 0.0;
#pragma unroll
        for (int j = 0; j < K1; j++) {
#pragma unroll 1
            for (int dg = 0; dg < PS::l; dg++) {
                const uint32_t pos = 32 - (dg + 1) * PS::Bgbit;
                double x[R];
#pragma unroll
                for (int r = 0; r < R; r++) x[r] = (double)(int32_t)__builtin_amdgcn_sbfe(temp[j][r], pos, (uint32_t)PS::Bgbit);
                PO::forward(x, ctx);
#pragma unroll
                for (int r = 0; r < R; r++) x[r] = fpf::reduce(x[r]);
                // (k+1) l narrow products of reduced factors per sum: below 2^53 unreduced (PsDims: ROWS * after_mulmod(0.5001) < LIM_WIDE)
                const double2* row = (const double2*)d.trgsw_ntt + ((size_t)(slot * D::ROWS + j * PS::l + dg) * K1) * (N / 2) + lane;
#pragma unroll
                for (int o = 0; o < K1; o++)
#pragma unroll
                    for (int q = 0; q < R / 2; q++) {
                        const double2 b = row[(size_t)o * (N / 2) + q * 64];
                        A[o][2 * q] += fpf::mulmod(x[2 * q], b.x);
                        A[o][2 * q + 1] += fpf::mulmod(x[2 * q + 1], b.y);
                    }
            }
        }
        const int shl = limb * PS::limb_bits;
#pragma unroll
        for (int o = 0; o < K1; o++) {
#pragma unroll
            for (int r = 0; r < R; r++) A[o][r] = fpf::reduce(A[o][r]);
            PO::inverse(A[o], ctx);
#pragma unroll
            for (int r = 0; r < R; r++) resw[o][r] += ps_lift<PS>(A[o][r]) << shl;
        }
    }
#pragma unroll
    for (int j = 0; j < K1; j++)
#pragma unroll
        for (int r = 0; r < R; r++) d.res[j * N + lane + 64 * r] = resw[j][r];
}

// ----------------------------------------------------------------------------------------------
// Key switch lvl1 -> lvl0 with the linear pre-add fused (include/keyswitch_gpu.cuh:83-188), one
// workgroup (16 waves) per ciphertext: wave w takes a'_j for j in [w kN/16, (w+1) kN/16), lane L the output
// words L, L + 64, ...; rows straight from L2; the 16 partial sums are added through LDS.
// ksk: [kN][t][2^(basebit-1)][n + 1] u32, the reference's layout unpadded.
// ----------------------------------------------------------------------------------------------
template <class PS>
__global__ __launch_bounds__(kKsThreads) void keyswitch_ps_kernel(
    const LinDesc* __restrict__ descs, int count, const uint32_t* __restrict__ ksk)
{
    using D = PsDims<PS>;
    constexpr int KN = PS::k * D::N, W0 = D::lvl0_words, PER = (W0 + 63) / 64;
    __shared__ uint32_t part[kKsWaves][PER * 64];
    __shared__ uint16_t dig[KN];
    __shared__ uint32_t bprime_s;
    static_assert(PS::t * PS::basebit <= 16, "digit word is 16 bits");
    const int g = blockIdx.x;
    if (g >= count) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const LinDesc d = descs[g];
    uint32_t koff = (PS::t * PS::basebit < 32) ? 1u << (32 - (1 + PS::basebit * PS::t)) : 0u;     // roundoffset, :92-98
    for (int i = 1; i <= PS::t; i++) koff += ((1u << PS::basebit) / 2) << (32 - i * PS::basebit); // iksoffsetgen, :13-23
    for (int j = tid; j <= KN; j += kKsThreads) {
        const uint32_t v = (uint32_t)d.ca * d.in0[j] + (uint32_t)d.cb * d.in1[j];
        if (j == KN) bprime_s = v + d.off;
        else dig[j] = (uint16_t)((v + koff) >> 16);
    }
    __syncthreads();
    uint32_t res[PER];
#pragma unroll
    for (int m = 0; m < PER; m++) res[m] = 0;
    constexpr int JW = KN / kKsWaves;
#pragma unroll 1
    for (int jj = 0; jj < JW; jj++) {
        const int j = wave * JW + jj;
        const uint32_t dj = __builtin_amdgcn_readfirstlane((uint32_t)dig[j]);
#pragma unroll
        for (int kap = 0; kap < PS::t; kap++) {
            const int val = (int)((dj >> (16 - (kap + 1) * PS::basebit)) & ((1u << PS::basebit) - 1)) - (1 << (PS::basebit - 1));
            if (val == 0) continue;       // wave-uniform
            const int v = val > 0 ? val : -val;
            const uint32_t* row = ksk + (((size_t)j * PS::t + kap) * D::ks_numbase + (v - 1)) * W0;
#pragma unroll
            for (int m = 0; m < PER; m++) {
                const int i = lane + 64 * m;
                const uint32_t w = i < W0 ? row[i] : 0u;
                res[m] = val > 0 ? res[m] - w : res[m] + w;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < PER; m++) part[wave][lane + 64 * m] = res[m];
    __syncthreads();
    for (int i = tid; i < W0; i += kKsThreads) {
        uint32_t v = (i == PS::n) ? bprime_s : 0u;
#pragma unroll
        for (int w = 0; w < kKsWaves; w++) v += part[w][i];
        d.out[i] = v;
    }
}

// ----------------------------------------------------------------------------------------------
// The same key switch with the table shared through LDS: keyswitch_kernel (kernels.hip.h) over the set's shape.
// ksk_padded: [kN][t][2^(basebit-1)][row_pad] u32, rows padded to a multiple of 128 words (paramsets.inc.h).
// ----------------------------------------------------------------------------------------------
template <class PS>
struct PsKs {
    using D = PsDims<PS>;
    static constexpr int KN = PS::k * D::N;
    static constexpr int W0 = D::lvl0_words;
    static constexpr int row_pad = (W0 + 127) / 128 * 128;                 // words: two quads per lane and whole pairs behind them
    static constexpr int step_rows = PS::t * D::ks_numbase;
    static_assert(PS::basebit == kKsBasebit, "keyswitch_kernel decodes digits of two bits (values -2 .. 1)");
};
template <class PS>
struct KsShapePs {
    using Desc = LinDesc;
    static constexpr int kn = PsKs<PS>::KN, t = PS::t, row_pad = PsKs<PS>::row_pad, n_out = PS::n;
    static constexpr uint32_t koff()
    {
        uint32_t o = (PS::t * PS::basebit < 32) ? 1u << (32 - (1 + PS::basebit * PS::t)) : 0u;     // roundoffset, keyswitch_gpu.cuh:92-98
        for (int i = 1; i <= PS::t; i++) o += ((1u << PS::basebit) / 2) << (32 - i * PS::basebit); // iksoffsetgen, :13-23
        return o;
    }
    static __device__ __forceinline__ uint32_t digit_word(const Desc& d, int j)
    {
        return ((uint32_t)d.ca * d.in0[j] + (uint32_t)d.cb * d.in1[j] + koff()) >> 16;
    }
    static __device__ __forceinline__ uint32_t bprime(const Desc& d) { return (uint32_t)d.ca * d.in0[kn] + (uint32_t)d.cb * d.in1[kn] + d.off; }
};

}  // namespace cufhe_amd
