// bench_api.cpp -- PCIe-inclusive rate of the reference-style per-gate API (not the headline metric).
// 4096 cufhe::Nand(out, a, b, st) calls on host-resident ciphertexts over 256 streams, then
// Synchronize(): what test/test_util.h:29-72 times in the reference ("Throughput: ms/gate").
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

#include "../include/cufhe_amd.hpp"

using namespace cufhe;
using P = TFHEpp::lvl0param;

int main(int argc, char** argv)
{
    const int kNumTests = argc > 1 ? atoi(argv[1]) : 4096, kNumStreams = 256;
    std::mt19937 eng(1);
    std::vector<uint32_t> bk((size_t)630 * 6 * 2 * 1024), ksk((size_t)1024 * 8 * 2 * 631);
    for (auto& v : bk) v = eng();
    for (auto& v : ksk) v = eng();
    SetGPUNum(1);
    Initialize(bk.data(), bk.size(), ksk.data(), ksk.size());
    std::vector<Ctxt<P>> a(kNumTests), b(kNumTests), o(kNumTests);
    for (int i = 0; i < kNumTests; i++)
        for (auto* c : {&a[i], &b[i]})
            for (auto& w : c->tlwehost) w = eng();
    std::vector<Stream> st(kNumStreams);
    for (auto& s : st) s.Create();
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < kNumTests; i++) Nand(o[i], a[i], b[i], st[i % kNumStreams]);
        auto t1 = std::chrono::steady_clock::now();
        Synchronize();
        auto t2 = std::chrono::steady_clock::now();
        const double enq = std::chrono::duration<double, std::milli>(t1 - t0).count();
        const double tot = std::chrono::duration<double, std::milli>(t2 - t0).count();
        std::printf("rep %d: %d Nand via per-gate API: enqueue %.2f ms, total %.2f ms, %.0f gates/s, %.4f ms/gate\n", rep,
                    kNumTests, enq, tot, kNumTests / (tot * 1e-3), tot / kNumTests);
    }
    for (auto& s : st) s.Destroy();
    CleanUp();
    return 0;
}
