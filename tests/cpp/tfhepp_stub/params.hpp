// tests/cpp/tfhepp_stub/params.hpp -- TEST-ONLY stand-in for TFHEpp's <params.hpp>.
//
// TFHEpp is an empty submodule in the reference tree (/root/reference/.gitmodules:1-3) and cannot be fetched here.  This file pins
// NOTHING about TFHEpp: it only declares, with the member names the reference's gate path uses (include/cufhe_gpu.cuh:40,60,102-146,
// src/cufhe_gates_gpu.cu:45-46, include/keyswitch_gpu.cuh:83-134), enough of its parameter structs and containers for
// include/cufhe_amd.hpp to be compiled with -DCUFHE_AMD_USE_TFHEPP -- the branch a real cuFHE user compiles -- so that branch cannot
// rot unseen.  Never shipped, never included by the library.
#pragma once
#include <array>
#include <cstdint>

// Like TFHEpp, the stand-in takes its numbers from the macro the build defines (CMakeLists.txt:8-24 of the reference passes
// USE_80BIT_SECURITY / USE_CONCRETE ... on to TFHEpp): the two alternative shapes the library has compiled sets for, and -- for the
// tests that a build on numbers NO compiled set has is refused -- single numbers overridden with -DTFHEPP_STUB_BGBIT / _T / _BASEBIT.
#if defined(USE_80BIT_SECURITY)
#define TFHEPP_STUB_N0 500
#define TFHEPP_STUB_NBIT 10
#define TFHEPP_STUB_K 1
#define TFHEPP_STUB_L 2
#define TFHEPP_STUB_BGBIT_DEFAULT 10
#elif defined(USE_CONCRETE)
#define TFHEPP_STUB_N0 630
#define TFHEPP_STUB_NBIT 9
#define TFHEPP_STUB_K 2
#define TFHEPP_STUB_L 3
#define TFHEPP_STUB_BGBIT_DEFAULT 6
#else
#define TFHEPP_STUB_N0 630
#define TFHEPP_STUB_NBIT 10
#define TFHEPP_STUB_K 1
#define TFHEPP_STUB_L 3
#define TFHEPP_STUB_BGBIT_DEFAULT 6
#endif
#ifndef TFHEPP_STUB_BGBIT
#define TFHEPP_STUB_BGBIT TFHEPP_STUB_BGBIT_DEFAULT
#endif
#ifndef TFHEPP_STUB_T
#define TFHEPP_STUB_T 8
#endif
#ifndef TFHEPP_STUB_BASEBIT
#define TFHEPP_STUB_BASEBIT 2
#endif

namespace TFHEpp {
struct lvl0param {
    using T = uint32_t;
    static constexpr uint32_t n = TFHEPP_STUB_N0, k = 1;
    static constexpr T mu = 1u << 29;
    static constexpr T μ = mu;
};
struct lvl1param {
    using T = uint32_t;
    static constexpr uint32_t nbit = TFHEPP_STUB_NBIT, n = 1u << nbit, k = TFHEPP_STUB_K, l = TFHEPP_STUB_L, Bgbit = TFHEPP_STUB_BGBIT, Bg = 1u << Bgbit;
    static constexpr T mu = 1u << 29;
    static constexpr T μ = mu;
};
struct lvl2param {
    using T = uint64_t;
    static constexpr uint32_t nbit = 11, n = 1u << nbit, k = 1, l = 4, Bgbit = 9, Bg = 1u << Bgbit;
    static constexpr T mu = 1ull << 61;
    static constexpr T μ = mu;
};
struct lvl01param { using domainP = lvl0param; using targetP = lvl1param; };
struct lvl02param { using domainP = lvl0param; using targetP = lvl2param; };
struct lvl10param {
    using domainP = lvl1param; using targetP = lvl0param;
    static constexpr uint32_t t = TFHEPP_STUB_T, basebit = TFHEPP_STUB_BASEBIT;
};
struct lvl20param {
    using domainP = lvl2param; using targetP = lvl0param;
    static constexpr uint32_t t = 7, basebit = 2;
};
template <class P> using Key = std::array<typename P::T, P::k * P::n>;
template <class P> using TLWE = std::array<typename P::T, P::k * P::n + 1>;
template <class P> using Polynomial = std::array<typename P::T, P::n>;
template <class P> using TRLWE = std::array<Polynomial<P>, P::k + 1>;
template <class P> using TRGSW = std::array<TRLWE<P>, (P::k + 1) * P::l>;
template <class P> using TRGSWNTT = std::array<std::array<std::array<uint64_t, P::n>, P::k + 1>, (P::k + 1) * P::l>;
template <class P> using BootstrappingKey = std::array<TRGSW<typename P::targetP>, P::domainP::k * P::domainP::n>;
template <class P>
using KeySwitchingKey = std::array<std::array<std::array<TLWE<typename P::targetP>, (1u << (P::basebit - 1))>, P::t>, P::domainP::k * P::domainP::n>;
}  // namespace TFHEpp
