// tests/cpp/tfhepp_stub/cloudkey.hpp -- TEST-ONLY stand-in for TFHEpp's <cloudkey.hpp> (see params.hpp beside it): struct EvalKey with
// the two accessors the reference's Initialize(ek) calls, ek.getbk<lvl01param>() and ek.getiksk<lvl10param>()
// (src/cufhe_gates_gpu.cu:42-47), plus the lvl02 / lvl20 pair include/cufhe_amd.hpp's lvl2::Initialize(ek) reads.
#pragma once
#include <memory>
#include <type_traits>

#include "params.hpp"

namespace TFHEpp {
struct EvalKey {
    std::unique_ptr<BootstrappingKey<lvl01param>> bklvl01;
    std::unique_ptr<BootstrappingKey<lvl02param>> bklvl02;
    std::unique_ptr<KeySwitchingKey<lvl10param>> iksklvl10;
    std::unique_ptr<KeySwitchingKey<lvl20param>> iksklvl20;
    template <class P> const BootstrappingKey<P>& getbk() const
    {
        if constexpr (std::is_same_v<P, lvl01param>) return *bklvl01;
        else {
            static_assert(std::is_same_v<P, lvl02param>, "stub: lvl01param or lvl02param");
            return *bklvl02;
        }
    }
    template <class P> const KeySwitchingKey<P>& getiksk() const
    {
        if constexpr (std::is_same_v<P, lvl10param>) return *iksklvl10;
        else {
            static_assert(std::is_same_v<P, lvl20param>, "stub: lvl10param or lvl20param");
            return *iksklvl20;
        }
    }
};
}  // namespace TFHEpp
