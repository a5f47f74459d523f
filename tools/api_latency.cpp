// api_latency.cpp -- wall time of ONE cufhe::Nand on host-resident ciphertexts through the per-gate API
// (record -> launch thread -> H2D -> kernels -> D2H -> tlwehost), and of a chain of dependent gates.
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

#include "../include/cufhe_amd.hpp"

using namespace cufhe;
using P = TFHEpp::lvl0param;

int main()
{
    std::mt19937 eng(1);
    std::vector<uint32_t> bk((size_t)630 * 6 * 2 * 1024), ksk((size_t)1024 * 8 * 2 * 631);
    for (auto& v : bk) v = eng();
    for (auto& v : ksk) v = eng();
    SetGPUNum(1);
    Initialize(bk.data(), bk.size(), ksk.data(), ksk.size());
    Ctxt<P> a, b, o;
    for (auto* c : {&a, &b})
        for (auto& w : c->tlwehost) w = eng();
    Stream st;
    st.Create();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto t0, auto t1) { return std::chrono::duration<double, std::milli>(t1 - t0).count(); };
    for (int rep = 0; rep < 6; rep++) {
        auto t0 = now();
        Nand(o, a, b, st);
        auto t1 = now();
        Synchronize();
        auto t2 = now();
        std::printf("one Nand: call %.3f ms, call + Synchronize %.3f ms\n", ms(t0, t1), ms(t0, t2));
    }
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = now();
        for (int k = 0; k < 16; k++) Nand(o, o, b, st);      // 16 dependent gates
        Synchronize();
        std::printf("chain of 16 dependent Nand: %.3f ms = %.3f ms per gate\n", ms(t0, now()), ms(t0, now()) / 16);
    }
    st.Destroy();
    CleanUp();
    return 0;
}
