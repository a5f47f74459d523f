// test_gate_api.cpp -- the reference's gate tests, written against include/cufhe_amd.hpp.
//
// Follows /root/reference/test/test_gate_gpu.cc:36-91 + test/test_util.h:8-95 (every gate on
// many streams, decrypt == plain truth function, Ctxt<lvl1param>), test_gate_gpu_multi.cc
// (same for Ctxt<lvl0param>), test_intensive.cc:21-128 (StreamQuery poll-and-refill over
// shared inputs) and test_api_gpu.cu:140-159 (chained in-place gates on one stream).
// Keys, encryption and decryption come from the CPU oracle (test infrastructure); seeds are
// fixed where the reference uses std::random_device.
#include <cstdio>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

#ifdef CUFHE_AMD_TEST_HIP      // the caller's side of Stream::st(): plain HIP runtime calls on the raw handle (StreamOrdering below)
#include <hip/hip_runtime_api.h>
#endif

#include "../../include/cufhe_amd.hpp"
#include "../../oracle/tfhe_oracle.h"
#include "../../oracle/tfhe_oracle_lvl2.h"

using namespace cufhe;

static std::vector<uint32_t> g_s0(ORC_n), g_s1(ORC_K * ORC_N);
static orc_rng g_rng;
static int g_failures = 0;

// scheduler counters summed over the devices, reset
static cufhe_amd_sched_stats all_stats()
{
    cufhe_amd_sched_stats sum{};
    for (int d = 0; d < GetGPUNum(); d++) {
        cufhe_amd_sched_stats s;
        CUFHE_AMD_CHECK(cufhe_amd_sched_get_stats(d, &s, 1));
        sum.gates += s.gates; sum.launch_sequences += s.launch_sequences; sum.levels += s.levels; sum.groups += s.groups;
        sum.renames += s.renames; sum.home_copies += s.home_copies;
    }
    return sum;
}

template <class P> const uint32_t* key() { return detail::level_of<P>() ? g_s1.data() : g_s0.data(); }
template <class P> void encrypt(Ctxt<P>& c, int bit) { orc_tlwe_encrypt(&g_rng, detail::level_of<P>(), key<P>(), bit, c.tlwehost.data()); }
template <class P> int decrypt(Ctxt<P>& c) { return orc_tlwe_decrypt(detail::level_of<P>(), key<P>(), c.tlwehost.data()); }

// test/test_util.h Test<P>(): out = ct[i], inputs ct[i+k*kNumTests]
template <class P, class Func>
void Test(const char* type, int op, int arity, Func func, std::vector<Ctxt<P>>& ct, std::vector<uint8_t>& pt,
          Stream* st, int kNumTests, int kNumSMs, std::mt19937& eng)
{
    for (int i = 0; i < 4 * kNumTests; i++) {
        pt[i] = eng() & 1;
        encrypt(ct[i], pt[i]);
    }
    for (int i = 0; i < kNumTests; i++) func(i, st[i % kNumSMs]);
    Synchronize();
    int bad = 0;
    for (int i = 0; i < kNumTests; i++) {
        int exp = orc_truth(op, pt[i + kNumTests], arity > 1 ? pt[i + 2 * kNumTests] : 0, arity > 2 ? pt[i + 3 * kNumTests] : 0);
        if (decrypt(ct[i]) != exp) bad++;
    }
    std::printf("%-6s level %d: %s (%d/%d failures)\n", type, detail::level_of<P>(), bad ? "FAIL" : "PASS", bad, kNumTests);
    g_failures += bad;
}

template <class P>
void AllGates(int kNumSMs, int kNumTests, std::mt19937& eng)
{
    std::vector<uint8_t> pt(4 * kNumTests);
    std::vector<Ctxt<P>> ct(4 * kNumTests);
    Stream* st = new Stream[kNumSMs];
    for (int i = 0; i < kNumSMs; i++) st[i].Create();
    const int K = kNumTests;
#define T2(NAME, OP, FN) Test<P>(NAME, OP, 2, [&](int i, Stream s) { FN<P>(ct[i], ct[i + K], ct[i + 2 * K], s); }, ct, pt, st, K, kNumSMs, eng)
    T2("NAND", ORC_NAND, Nand); T2("OR", ORC_OR, Or); T2("ORYN", ORC_ORYN, OrYN); T2("ORNY", ORC_ORNY, OrNY);
    T2("AND", ORC_AND, And); T2("ANDYN", ORC_ANDYN, AndYN); T2("ANDNY", ORC_ANDNY, AndNY);
    T2("XOR", ORC_XOR, Xor); T2("XNOR", ORC_XNOR, Xnor); T2("NOR", ORC_NOR, Nor);
#undef T2
    Test<P>("MUX", ORC_MUX, 3, [&](int i, Stream s) { Mux<P>(ct[i], ct[i + K], ct[i + 2 * K], ct[i + 3 * K], s); }, ct, pt, st, K, kNumSMs, eng);
    Test<P>("NMUX", ORC_NMUX, 3, [&](int i, Stream s) { NMux<P>(ct[i], ct[i + K], ct[i + 2 * K], ct[i + 3 * K], s); }, ct, pt, st, K, kNumSMs, eng);
    Test<P>("NOT", ORC_NOT, 1, [&](int i, Stream s) { Not<P>(ct[i], ct[i + K], s); }, ct, pt, st, K, kNumSMs, eng);
    Test<P>("COPY", ORC_COPY, 1, [&](int i, Stream s) { Copy<P>(ct[i], ct[i + K], s); }, ct, pt, st, K, kNumSMs, eng);
    for (int i = 0; i < kNumSMs; i++) st[i].Destroy();
    delete[] st;
}

// test/test_api_gpu.cu:140-159: chained gates, output aliasing an input, one Synchronize
void Chained(std::mt19937& eng)
{
    using P = TFHEpp::lvl0param;
    const int kNumTests = 64, kNumSMs = 8;
    std::vector<Ctxt<P>> a(kNumTests), b(kNumTests), c(kNumTests);
    std::vector<uint8_t> pa(kNumTests), pb(kNumTests), pc(kNumTests);
    Stream* st = new Stream[kNumSMs];
    for (int i = 0; i < kNumSMs; i++) st[i].Create();
    for (int i = 0; i < kNumTests; i++) {
        pa[i] = eng() & 1; pb[i] = eng() & 1; pc[i] = eng() & 1;
        encrypt(a[i], pa[i]); encrypt(b[i], pb[i]); encrypt(c[i], pc[i]);
    }
    all_stats();
    for (int i = 0; i < kNumTests; i++) {
        Stream s = st[i % kNumSMs];
        Nand(a[i], a[i], b[i], s); pa[i] = 1 - pa[i] * pb[i];
        Or(a[i], a[i], b[i], s);   pa[i] = pa[i] | pb[i];
        Xor(a[i], a[i], c[i], s);  pa[i] = pa[i] ^ pc[i];
        Not(a[i], a[i], s);        pa[i] = 1 - pa[i];
        Mux(a[i], a[i], b[i], c[i], s); pa[i] = pa[i] ? pb[i] : pc[i];
    }
    Synchronize();
    int bad = 0;
    for (int i = 0; i < kNumTests; i++) bad += decrypt(a[i]) != pa[i];
    // five dependence levels, however the 64 chains were interleaved: not 320 one-gate launches
    cufhe_amd_sched_stats stats = all_stats();
    if (stats.launch_sequences > 6 * (uint64_t)GetGPUNum()) bad++;
    std::printf("chained in-place gates: %s (%d/%d failures, %llu launch sequences for %llu gates)\n", bad ? "FAIL" : "PASS", bad,
                kNumTests, (unsigned long long)stats.launch_sequences, (unsigned long long)stats.gates);
    g_failures += bad;
    for (int i = 0; i < kNumSMs; i++) st[i].Destroy();
    delete[] st;
}

// test/test_intensive.cc:21-128: poll StreamQuery and refill, inputs shared by all streams
void Intensive(std::mt19937& eng)
{
    using P = TFHEpp::lvl0param;
    const int kNumStreams = 1000, kRounds = 20;      // the reference's sizes (test/test_intensive.cc:21-24)
    Ctxt<P> in0, in1, inc;
    int p0 = eng() & 1, p1 = eng() & 1, pcb = eng() & 1;
    encrypt(in0, p0); encrypt(in1, p1); encrypt(inc, pcb);
    std::vector<Ctxt<P>> out(kNumStreams);
    std::vector<int> round(kNumStreams, 0);
    Stream* st = new Stream[kNumStreams];
    for (int i = 0; i < kNumStreams; i++) st[i].Create();
    int bad = 0, done = 0;
    for (int i = 0; i < kNumStreams; i++) Nand(out[i], in0, in1, st[i]);
    while (done < kNumStreams) {
        for (int i = 0; i < kNumStreams; i++) {
            if (round[i] >= kRounds || !StreamQuery(st[i])) continue;
            const bool was_nand = (round[i] % 2) == 0;
            int exp = was_nand ? 1 - p0 * p1 : (pcb ? p1 : p0);
            bad += decrypt(out[i]) != exp;
            if (++round[i] == kRounds) { done++; continue; }
            if (round[i] % 2) Mux(out[i], inc, in1, in0, st[i]);
            else Nand(out[i], in0, in1, st[i]);
        }
    }
    Synchronize();
    std::printf("intensive poll-and-refill: %s (%d failures over %d gates)\n", bad ? "FAIL" : "PASS", bad, kNumStreams * kRounds);
    g_failures += bad;
    for (int i = 0; i < kNumStreams; i++) st[i].Destroy();
    delete[] st;
}

// g-gates: device-resident chaining with explicit copies (include/cufhe_gpu.cuh:282-313)
void DeviceResident(std::mt19937& eng)
{
    using P = TFHEpp::lvl1param;
    const int K = 32;
    std::vector<Ctxt<P>> a(K), b(K), o(K);
    std::vector<uint8_t> pa(K), pb(K);
    Stream st;
    st.Create();
    for (int i = 0; i < K; i++) { pa[i] = eng() & 1; pb[i] = eng() & 1; encrypt(a[i], pa[i]); encrypt(b[i], pb[i]); }
    for (int i = 0; i < K; i++) {
        CtxtCopyH2D(a[i], st); CtxtCopyH2D(b[i], st);
        gAnd(o[i], a[i], b[i], st);
        gXor(o[i], o[i], b[i], st);
        CtxtCopyD2H(o[i], st);
    }
    Synchronize();
    int bad = 0;
    for (int i = 0; i < K; i++) bad += decrypt(o[i]) != ((pa[i] & pb[i]) ^ pb[i]);
    std::printf("g-gates with explicit copies: %s (%d/%d failures)\n", bad ? "FAIL" : "PASS", bad, K);
    g_failures += bad;
    st.Destroy();
}

// What Stream::st() means.  In the reference a gate IS enqueued on st.st() at the call (src/cufhe_gates_gpu.cu:148-167,
// include/cufhe_gpu.cuh:183), so a caller may put its own work on that stream and rely on stream order.  The idioms, word for word
// against the oracle's gate:
//   (a) gNand(out, ..); StreamSynchronize(st);                    then a plain copy out of out.tlwedevices[d]
//   (b) gNand(out, ..); hipMemcpyAsync(host, out.tlwedevices[d], .., st.st()); hipStreamSynchronize(st.st())
//   (c) hipMemcpyAsync(in.tlwedevices[d], words, .., st.st()); gNand(out, in, ..)   -- the gate runs behind the caller's upload
//   (d) hipEventRecord(ev, st.st()) after a gate; hipEventSynchronize(ev)         -- the event completes behind the gate
//   (e) Nand(out, ..); StreamSynchronize(st)                      tlwehost holds the result (cudaStreamSynchronize(st.st()) in the reference)
void StreamOrdering(std::mt19937& eng)
{
    using P = TFHEpp::lvl0param;
    // (the oracle's evaluation key refers to the key-switching key it was given: the arrays live as long as it does)
    std::vector<uint32_t> obk(ORC_BK_WORDS), oksk(ORC_KSK_WORDS);
    orc_bkgen(1001, g_s0.data(), g_s1.data(), obk.data());
    orc_kskgen(2001, g_s0.data(), g_s1.data(), oksk.data());
    orc_evalkey* ek = orc_evalkey_create(obk.data(), oksk.data());
    int bad = 0, total = 0;
    Stream st;
    st.Create();
    const int d = st.device_id();
    constexpr size_t kBytes = sizeof(TFHEpp::TLWE<P>);
    auto want = [&](int op, const TFHEpp::TLWE<P>& x, const TFHEpp::TLWE<P>& y) {
        TFHEpp::TLWE<P> w{};
        orc_gate(ek, op, 0, w.data(), x.data(), y.data(), nullptr);
        return w;
    };
    for (int rep = 0; rep < 3; rep++) {
        Ctxt<P> a, b, out, out2, out3, out4, out5;
        encrypt(a, eng() & 1); encrypt(b, eng() & 1);
        CtxtCopyH2D(a, st); CtxtCopyH2D(b, st);
        // (a)
        gNand(out, a, b, st);
        StreamSynchronize(st);
        TFHEpp::TLWE<P> got{};
        CUFHE_AMD_CHECK(cufhe_amd_memcpy_d2h(d, nullptr, got.data(), out.tlwedevices[d], kBytes));
        CUFHE_AMD_CHECK(cufhe_amd_stream_synchronize(d, nullptr));
        bad += got != want(ORC_NAND, a.tlwehost, b.tlwehost); total++;
#ifdef CUFHE_AMD_TEST_HIP
        // (b): a chain of two recorded gates, then the caller's own copy and wait on the raw handle
        gXor(out2, out, b, st);
        gAnd(out3, out2, a, st);
        TFHEpp::TLWE<P> got2{}, got3{};
        hipStream_t raw = st.st();
        bad += hipMemcpyAsync(got3.data(), out3.tlwedevices[d], kBytes, hipMemcpyDeviceToHost, raw) != hipSuccess;
        bad += hipMemcpyAsync(got2.data(), out2.tlwedevices[d], kBytes, hipMemcpyDeviceToHost, raw) != hipSuccess;
        bad += hipStreamSynchronize(raw) != hipSuccess;
        const auto w2 = want(ORC_XOR, got, b.tlwehost), w3 = want(ORC_AND, w2, a.tlwehost);
        bad += got2 != w2; total++;
        bad += got3 != w3; total++;
        // (c): the caller uploads fresh words into a ciphertext's own device buffer through the handle, then a g-gate reads it
        TFHEpp::TLWE<P> fresh{};
        orc_tlwe_encrypt(&g_rng, 0, g_s0.data(), (int)(eng() & 1), fresh.data());
        bad += hipMemcpyAsync(out2.tlwedevices[d], fresh.data(), kBytes, hipMemcpyHostToDevice, st.st()) != hipSuccess;
        gOr(out4, out2, b, st);
        // (d): an event behind the gate
        hipEvent_t ev;
        bad += hipEventCreate(&ev) != hipSuccess;
        bad += hipEventRecord(ev, st.st()) != hipSuccess;
        bad += hipEventSynchronize(ev) != hipSuccess;
        TFHEpp::TLWE<P> got4{};
        bad += hipMemcpy(got4.data(), out4.tlwedevices[d], kBytes, hipMemcpyDeviceToHost) != hipSuccess;
        bad += got4 != want(ORC_OR, fresh, b.tlwehost); total++;
        (void)hipEventDestroy(ev);
#endif
        // (e)
        Nand(out5, a, b, st);
        StreamSynchronize(st);
        bad += out5.tlwehost != want(ORC_NAND, a.tlwehost, b.tlwehost); total++;
    }
    std::printf("Stream::st() ordering (StreamSynchronize, copies / events / uploads on the raw handle): %s (%d/%d failures)\n", bad ? "FAIL" : "PASS", bad, total);
    g_failures += bad;
    st.Destroy();
    orc_evalkey_destroy(ek);
}

// A real circuit through the per-gate API: 16 independent 8-bit ripple-carry adders, one per
// stream, every gate depending on earlier ones (the scheduler must cut the recorded gates into
// dependence levels and still batch across the 16 adders).
void RippleAdders(std::mt19937& eng)
{
    using P = TFHEpp::lvl0param;
    const int kAdders = 16, kBits = 8;
    std::vector<Ctxt<P>> a(kAdders * kBits), b(kAdders * kBits), sum(kAdders * kBits), carry(kAdders), t1(kAdders), t2(kAdders);
    std::vector<unsigned> va(kAdders), vb(kAdders);
    Stream* st = new Stream[kAdders];
    for (int i = 0; i < kAdders; i++) st[i].Create();
    for (int i = 0; i < kAdders; i++) {
        va[i] = eng() & 0xff; vb[i] = eng() & 0xff;
        for (int k = 0; k < kBits; k++) { encrypt(a[i * kBits + k], (va[i] >> k) & 1); encrypt(b[i * kBits + k], (vb[i] >> k) & 1); }
        encrypt(carry[i], 0);
    }
    all_stats();
    for (int k = 0; k < kBits; k++)
        for (int i = 0; i < kAdders; i++) {
            Ctxt<P>&x = a[i * kBits + k], &y = b[i * kBits + k], &s = sum[i * kBits + k], &c = carry[i];
            Xor(t1[i], x, y, st[i]);          // t1 = x ^ y
            Xor(s, t1[i], c, st[i]);          // s  = t1 ^ c
            And(t2[i], t1[i], c, st[i]);      // t2 = t1 & c
            And(t1[i], x, y, st[i]);          // t1 = x & y       (overwrites t1 after its readers)
            Or(c, t1[i], t2[i], st[i]);       // c  = t1 | t2     (in place on the carry)
        }
    Synchronize();
    int bad = 0;
    for (int i = 0; i < kAdders; i++) {
        unsigned got = 0;
        for (int k = 0; k < kBits; k++) got |= (unsigned)decrypt(sum[i * kBits + k]) << k;
        got |= (unsigned)decrypt(carry[i]) << kBits;
        bad += got != va[i] + vb[i];
    }
    cufhe_amd_sched_stats stats = all_stats();
    // renaming (the default): the re-used temporaries no longer order the program, two levels per bit + the level of copies home
    const bool renaming = !getenv("CUFHE_AMD_NO_SCHED_RENAME");
    if (stats.launch_sequences > (renaming ? 2 * kBits + 2 : 40) * (uint64_t)GetGPUNum()) bad++;
    if ((stats.renames > 0) != renaming) bad++;
    // Ctxt::tlwedevices (include/cufhe_gpu.cuh:80-84) is a public member: after Synchronize() the buffer it names holds the
    // ciphertext, whether or not the scheduler kept the value somewhere else on the way
    int stale = 0;
    for (int i = 0; i < kAdders; i++)
        for (Ctxt<P>* c : {&t1[i], &t2[i], &carry[i]}) {
            TFHEpp::TLWE<P> dev{};
            const int d = st[i].device_id();
            CUFHE_AMD_CHECK(cufhe_amd_memcpy_d2h(d, nullptr, dev.data(), c->tlwedevices[d], sizeof(dev)));
            CUFHE_AMD_CHECK(cufhe_amd_stream_synchronize(d, nullptr));
            stale += dev != c->tlwehost;
        }
    bad += stale;
    std::printf("16 x 8-bit ripple-carry adders (640 dependent gates): %s (%d/%d wrong sums, %llu launch sequences, %llu renames, %d stale tlwedevices)\n",
                bad ? "FAIL" : "PASS", bad, kAdders, (unsigned long long)stats.launch_sequences, (unsigned long long)stats.renames, stale);
    g_failures += bad;
    for (int i = 0; i < kAdders; i++) st[i].Destroy();
    delete[] st;
}

// BASELINE configs[2] at full size on one GPU, as the reference's harness runs it (test/test_util.h:29-94):
// 32 768 mixed AND / OR / XOR / NAND gates round-robin over 256 streams, Synchronize, decrypt all.
void MixedAtSize(std::mt19937& eng)
{
    using P = TFHEpp::lvl0param;
    const int kGates = 32768, kStreams = 256;
    std::vector<Ctxt<P>> a(kGates), b(kGates), o(kGates);
    std::vector<uint8_t> pa(kGates), pb(kGates);
    Stream* st = new Stream[kStreams];
    for (int i = 0; i < kStreams; i++) st[i].Create();
    for (int i = 0; i < kGates; i++) { pa[i] = eng() & 1; pb[i] = eng() & 1; encrypt(a[i], pa[i]); encrypt(b[i], pb[i]); }
    for (int i = 0; i < kGates; i++) {
        Stream s = st[i % kStreams];
        switch (i % 4) {
            case 0: And(o[i], a[i], b[i], s); break;
            case 1: Or(o[i], a[i], b[i], s); break;
            case 2: Xor(o[i], a[i], b[i], s); break;
            default: Nand(o[i], a[i], b[i], s); break;
        }
    }
    Synchronize();
    int bad = 0;
    for (int i = 0; i < kGates; i++) {
        const int exp = i % 4 == 0 ? (pa[i] & pb[i]) : i % 4 == 1 ? (pa[i] | pb[i]) : i % 4 == 2 ? (pa[i] ^ pb[i]) : 1 - (pa[i] & pb[i]);
        bad += decrypt(o[i]) != exp;
    }
    std::printf("32768 mixed gates on 256 streams: %s (%d failures)\n", bad ? "FAIL" : "PASS", bad);
    g_failures += bad;
    for (int i = 0; i < kStreams; i++) st[i].Destroy();
    delete[] st;
}

// test/test_perf.cc:36-87 on whatever parameter set this build runs on (GateBootstrappingTLWE2TRLWElvl01NTT then Refresh, decrypt
// coefficient 0), plus SampleExtractAndKeySwitch: the three TRLWE-level operations every build of the reference has -- the
// small-modulus one included (src/cufhe_gates_gpu.cu:86-146 are outside its #ifndef)
void TrlweBootstraps(std::mt19937& eng)
{
    using namespace TFHEpp;
    const int kNum = 12, kStreams = 3;
    std::vector<Stream> st(kStreams);
    for (auto& s : st) s.Create();
    std::vector<Ctxt<lvl0param>> in(kNum), out0(kNum);
    std::vector<cuFHETRLWElvl1> t(kNum), r(kNum);
    std::vector<int> bits(kNum);
    for (int i = 0; i < kNum; i++) {
        bits[i] = eng() & 1;
        encrypt(in[i], bits[i]);
        GateBootstrappingTLWE2TRLWElvl01NTT(t[i], in[i], st[i % kStreams]);
        Refresh(r[i], t[i], st[i % kStreams]);
        SampleExtractAndKeySwitch(out0[i], r[i], st[i % kStreams]);
    }
    Synchronize();
    int bad = 0;
    auto coeff0 = [&](cuFHETRLWElvl1& x) {
        uint32_t tl[ORC_LVL1_WORDS];
        orc_sample_extract0(tl, x.trlwehost[0].data());
        return orc_tlwe_decrypt(1, g_s1.data(), tl);
    };
    for (int i = 0; i < kNum; i++) bad += (coeff0(t[i]) != bits[i]) + (coeff0(r[i]) != bits[i]) + (decrypt(out0[i]) != bits[i]);
    std::printf("bootstrap to TRLWE, Refresh, SampleExtractAndKeySwitch on %d ciphertexts: %s (%d failures)\n", kNum, bad ? "FAIL" : "PASS", bad);
    g_failures += bad;
    for (auto& s : st) s.Destroy();
}

#ifndef CUFHE_AMD_SMALL_NTT_MODULUS
// test/test_perf.cc:36-87 (GateBootstrappingTLWE2TRLWElvl01NTT then Refresh, decrypt coefficient 0)
// and test/test_cmux.cc:36-150 (CMUXNTT on TRLWE/TRGSW), plus SampleExtractAndKeySwitch -- on whatever parameter set this build runs
// on: in the reference the set TFHEpp selects serves CMUXNTT / TRGSW2NTT too (src/bootstrap_gpu.cu:75-94,197-285); only its
// small-modulus build has none (src/cufhe_gates_gpu.cu:68-86).
void TrlwePrimitives(std::mt19937& eng, const std::vector<uint32_t>& bk)
{
    using namespace TFHEpp;
    Stream st;
    st.Create();
    int bad = 0, total = 0;
    auto coeff0 = [&](cuFHETRLWElvl1& t) {
        uint32_t tl[ORC_LVL1_WORDS];
        orc_sample_extract0(tl, t.trlwehost[0].data());
        return orc_tlwe_decrypt(1, g_s1.data(), tl);
    };
    for (int rep = 0; rep < 8; rep++) {
        const int bit = eng() & 1;
        Ctxt<lvl0param> in, out0;
        encrypt(in, bit);
        cuFHETRLWElvl1 t, r;
        GateBootstrappingTLWE2TRLWElvl01NTT(t, in, st);
        Refresh(r, t, st);
        SampleExtractAndKeySwitch(out0, r, st);
        Synchronize();                       // recorded like gates: results are in the host members now (test/test_perf.cc:81)
        bad += coeff0(t) != bit; total++;
        bad += coeff0(r) != bit; total++;
        bad += decrypt(out0) != bit; total++;
        // CMUX: the bootstrapping key row i is a TRGSW encryption of s0[i]
        const int i = eng() % ORC_n;
        TRGSW<lvl1param> trgsw;
        static_assert(sizeof(trgsw) == (size_t)ORC_BK_ROWS * (ORC_K + 1) * ORC_N * sizeof(uint32_t), "TRGSW<lvl1param> is one step of the oracle's key");
        std::memcpy(trgsw.data(), bk.data() + (size_t)i * ORC_BK_ROWS * (ORC_K + 1) * ORC_N, sizeof(trgsw));
        cuFHETRGSWNTTlvl1 cs;
        TRGSW2NTT(cs, trgsw, st);
        Ctxt<lvl0param> other;
        encrypt(other, 1 - bit);
        cuFHETRLWElvl1 t_other, res, res2;
        GateBootstrappingTLWE2TRLWElvl01NTT(t_other, other, st);
        // recorded like the reference's (src/cufhe_gates_gpu.cu:68-85: asynchronous on st): t_other is still on its way
        // to its trlwehost when CMUXNTT is called -- the scheduler orders the CMUX behind the bootstrap, no Synchronize
        CMUXNTT(res, cs, t, t_other, st);          // s0[i] ? t : t_other
        CMUXNTT(res2, cs, t_other, res, st);       // chained on a recorded result, operands swapped: s0[i] ? t_other : res
        Synchronize();
        bad += coeff0(res) != (g_s0[i] ? bit : 1 - bit); total++;
        bad += coeff0(res2) != 1 - bit; total++;   // either branch holds 1 - bit
        // TRGSW2NTT leaves the words on the stream's device as well (src/bootstrap_gpu.cu:75-94): the device-resident form
        // must find them there (t, t_other are on the device since the operations above)
        cuFHETRLWElvl1 res3;
        gCMUXNTT(res3, cs, t, t_other, st);
        CUFHE_AMD_CHECK(cufhe_amd_enqueue_copy(st.device_id(), st.raw(), res3.handle, 0));
        // the holder refilled between two unsynchronised CMUXNTT calls: the first must use the old selector, the second the new
        const int i2 = (i + 1 + (int)(eng() % (ORC_n - 30))) % ORC_n;
        TRGSW<lvl1param> trgsw2;
        std::memcpy(trgsw2.data(), bk.data() + (size_t)i2 * ORC_BK_ROWS * (ORC_K + 1) * ORC_N, sizeof(trgsw2));
        cuFHETRLWElvl1 res4, res5;
        CMUXNTT(res4, cs, t, t_other, st);         // s0[i]  ? t : t_other
        TRGSW2NTT(cs, trgsw2, st);
        CMUXNTT(res5, cs, t, t_other, st);         // s0[i2] ? t : t_other
        Synchronize();
        bad += coeff0(res3) != (g_s0[i] ? bit : 1 - bit); total++;
        bad += coeff0(res4) != (g_s0[i] ? bit : 1 - bit); total++;
        bad += coeff0(res5) != (g_s0[i2] ? bit : 1 - bit); total++;
    }
    std::printf("TRLWE-level primitives: %s (%d/%d failures)\n", bad ? "FAIL" : "PASS", bad, total);
    g_failures += bad;
    st.Destroy();
}
#endif  // CUFHE_AMD_SMALL_NTT_MODULUS

#ifdef ORC_SET_DEFAULT
// test/test_perf.cc:36-87 at size: 4096 x GateBootstrappingTLWE2TRLWElvl01NTT, then 4096 x Refresh on 800 streams
void RefreshAtSize(std::mt19937& eng)
{
    using namespace TFHEpp;
    const int kNum = 4096, kStreams = 800;
    std::vector<Ctxt<lvl0param>> in(kNum);
    std::vector<cuFHETRLWElvl1> t(kNum), r(kNum);
    std::vector<uint8_t> bits(kNum);
    std::vector<Stream> st(kStreams);
    for (auto& s : st) s.Create();
    for (int i = 0; i < kNum; i++) { bits[i] = eng() & 1; encrypt(in[i], bits[i]); }
    all_stats();
    for (int i = 0; i < kNum; i++) GateBootstrappingTLWE2TRLWElvl01NTT(t[i], in[i], st[i % kStreams]);
    for (int i = 0; i < kNum; i++) Refresh(r[i], t[i], st[i % kStreams]);
    Synchronize();
    cufhe_amd_sched_stats stats = all_stats();
    int bad = 0;
    for (int i = 0; i < kNum; i++) {
        uint32_t tl[ORC_LVL1_WORDS];
        orc_sample_extract0(tl, r[i].trlwehost[0].data());
        bad += orc_tlwe_decrypt(1, g_s1.data(), tl) != bits[i];
    }
    if (stats.launch_sequences > 8 * (uint64_t)GetGPUNum()) bad++;
    std::printf("4096 bootstraps to TRLWE + 4096 Refresh on 800 streams: %s (%d failures, %llu launch sequences)\n", bad ? "FAIL" : "PASS",
                bad, (unsigned long long)stats.launch_sequences);
    g_failures += bad;
    for (auto& s : st) s.Destroy();
}

// BASELINE configs[4]: the same gates through the N = 2048 ring (no reference test exists:
// the check is the one of test/test_util.h:75-94, decrypt == plain truth function)
void Lvl2Gates(std::mt19937& eng)
{
    std::vector<uint32_t> s2(ORC2_N);
    orc2_keygen(7, s2.data());
    std::vector<uint64_t> bk(ORC2_BK_WORDS);
    std::vector<uint32_t> ksk(ORC2_KSK_WORDS);
    orc2_bkgen(3007, g_s0.data(), s2.data(), bk.data());
    orc2_kskgen(4007, g_s0.data(), s2.data(), ksk.data());
#ifdef CUFHE_AMD_USE_TFHEPP
    {   // the form a cuFHE user with TFHEpp writes: the keys travel inside a TFHEpp::EvalKey
        TFHEpp::EvalKey ek2;
        ek2.bklvl02 = std::make_unique<TFHEpp::BootstrappingKey<TFHEpp::lvl02param>>();
        ek2.iksklvl20 = std::make_unique<TFHEpp::KeySwitchingKey<TFHEpp::lvl20param>>();
        static_assert(sizeof(*ek2.bklvl02) == ORC2_BK_WORDS * sizeof(uint64_t) && sizeof(*ek2.iksklvl20) == ORC2_KSK_WORDS * sizeof(uint32_t),
                      "stub containers and the oracle's key layouts have the same size");
        std::memcpy(ek2.bklvl02->data(), bk.data(), sizeof(*ek2.bklvl02));
        std::memcpy(ek2.iksklvl20->data(), ksk.data(), sizeof(*ek2.iksklvl20));
        lvl2::Initialize(ek2);
    }
#else
    lvl2::Initialize(bk.data(), bk.size(), ksk.data(), ksk.size());
#endif
    const int K = 64, W = ORC_LVL0_WORDS;
    std::vector<uint32_t> h(4 * K * W);
    std::vector<uint8_t> pt(3 * K);
    uint32_t* d = nullptr;
    CUFHE_AMD_CHECK(cufhe_amd_malloc(0, h.size() * 4, (void**)&d));
    const int ops[] = {ORC_NAND, ORC_XOR, ORC_ORNY, ORC_MUX, ORC_NMUX, ORC_NOT};
    for (int op : ops) {
        for (int i = 0; i < 3 * K; i++) {
            pt[i] = eng() & 1;
            orc_tlwe_encrypt(&g_rng, 0, g_s0.data(), pt[i], h.data() + (size_t)(K + i) * W);
        }
        CUFHE_AMD_CHECK(cufhe_amd_memcpy_h2d(0, nullptr, d, h.data(), h.size() * 4));
        lvl2::GateBatch(op, K, d, d + K * W, d + 2 * K * W, d + 3 * K * W);
        CUFHE_AMD_CHECK(cufhe_amd_memcpy_d2h(0, nullptr, h.data(), d, (size_t)K * W * 4));
        Synchronize();
        int bad = 0;
        for (int i = 0; i < K; i++)
            if (orc_tlwe_decrypt(0, g_s0.data(), h.data() + (size_t)i * W) != orc_truth(op, pt[i], pt[K + i], pt[2 * K + i])) bad++;
        std::printf("lvl2 ring op %d: %s (%d/%d failures)\n", op, bad ? "FAIL" : "PASS", bad, K);
        g_failures += bad;
    }
    CUFHE_AMD_CHECK(cufhe_amd_free(0, d));
    // the whole per-gate API (host ciphertexts, streams, scheduler) over the same ring
    CUFHE_AMD_CHECK(cufhe_amd_set_option("lvl0_ring", 2048));
    AllGates<TFHEpp::lvl0param>(16, 64, eng);
    CUFHE_AMD_CHECK(cufhe_amd_set_option("lvl0_ring", 1024));
}

#endif  // ORC_SET_DEFAULT

// Source written against the reference's header touches its public globals and the stream type directly
// (include/cufhe_gpu.cuh:44-46 `extern int _gpuNum; extern int streamCount;`, :154-165 the default Stream constructor,
// :183 `cudaStream_t st()`): the same lines must compile and behave here.
void ReferenceGlobals(int gpus)
{
    int bad = 0;
    bad += cufhe::_gpuNum != gpus;                          // set by SetGPUNum, src/cufhe_gates_gpu.cu:38
    const int before = cufhe::streamCount;
    Stream a, b(0);
    bad += cufhe::streamCount != before + 2;                // both constructors count, :154-165
    bad += a.device_id() != before % cufhe::_gpuNum;        // round-robin over the devices
    for (int i = 0; i < cufhe::_gpuNum; i++) bad += i >= GetGPUNum();        // the reference's `for (i < _gpuNum)` loops
    a.Create();
    cufheStream_t raw = a.st();                             // the type st() returns: hipStream_t without the HIP headers
    bad += raw == nullptr;
    bad += !StreamQuery(a);                                 // an idle stream
    a.Destroy();
    bad += a.st() != nullptr;
    std::printf("reference globals (_gpuNum, streamCount, cufheStream_t): %s\n", bad ? "FAIL" : "PASS");
    g_failures += bad;
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);      // a crash must not take the lines already printed with it
    const int gpus = argc > 1 ? atoi(argv[1]) : 1;
    const int kNumSMs = 64, kNumTests = kNumSMs * 4;
    std::mt19937 eng(12345);
    orc_rng_seed(&g_rng, 999);
    orc_keygen(1, g_s0.data(), g_s1.data());
    std::vector<uint32_t> bk(ORC_BK_WORDS), ksk(ORC_KSK_WORDS);
    orc_bkgen(1001, g_s0.data(), g_s1.data(), bk.data());
    orc_kskgen(2001, g_s0.data(), g_s1.data(), ksk.data());

    // more logical GPUs than the box has (test/test_gate_gpu_multi.cc hard-codes gpuNum = 2): every logical device
    // keeps its own keys, scheduler, launch thread and streams, several of them on one physical GPU
    if (getenv("CUFHE_AMD_SHARE_DEVICES")) CUFHE_AMD_CHECK(cufhe_amd_set_option("share_devices", 1));
    // the whole program again with every output waiting for the users of its buffer (include/cufhe_amd.h, "sched_rename" 0)
    if (getenv("CUFHE_AMD_NO_SCHED_RENAME")) CUFHE_AMD_CHECK(cufhe_amd_set_option("sched_rename", 0));
    SetGPUNum(gpus);
#ifdef CUFHE_AMD_USE_TFHEPP
    {   // Initialize(const TFHEpp::EvalKey&), src/cufhe_gates_gpu.cu:42-47 -- here over the test-only stand-in headers of
        // tests/cpp/tfhepp_stub (TFHEpp itself is not in the reference tree): the branch of cufhe_amd.hpp a cuFHE user compiles
        TFHEpp::EvalKey ek;
        ek.bklvl01 = std::make_unique<TFHEpp::BootstrappingKey<TFHEpp::lvl01param>>();
        ek.iksklvl10 = std::make_unique<TFHEpp::KeySwitchingKey<TFHEpp::lvl10param>>();
        static_assert(sizeof(*ek.bklvl01) == ORC_BK_WORDS * sizeof(uint32_t) && sizeof(*ek.iksklvl10) == ORC_KSK_WORDS * sizeof(uint32_t),
                      "stub containers and the oracle's key layouts have the same size");
        std::memcpy(ek.bklvl01->data(), bk.data(), sizeof(*ek.bklvl01));
        std::memcpy(ek.iksklvl10->data(), ksk.data(), sizeof(*ek.iksklvl10));
        Initialize(ek);
    }
#else
    Initialize(bk.data(), bk.size(), ksk.data(), ksk.size());
#endif
    ReferenceGlobals(gpus);
    AllGates<TFHEpp::lvl1param>(kNumSMs, kNumTests, eng);   // test_gate_gpu.cc
    AllGates<TFHEpp::lvl0param>(kNumSMs, kNumTests, eng);   // test_gate_gpu_multi.cc
#ifndef ORC_SET_DEFAULT
    // a build on another parameter set (TFHEpp's structs -- or the stand-ins selected by -DCUFHE_AMD_PARAM_SET_... -- with the oracle
    // compiled for the same set, -DORC_SET_...): the header FOUND the library's set from the structs' numbers.  The gate tests above ran
    // key switch -> blind rotate on lvl1 ciphertexts and blind rotate -> key switch on lvl0 ciphertexts of that set; then the programs
    // that only need gates and the TRLWE-level operations
    static_assert(kParamSetIndex > 0, "a non-default oracle build belongs to a non-default parameter set");
    Chained(eng);
    Intensive(eng);
    DeviceResident(eng);
    TrlweBootstraps(eng);
#ifndef CUFHE_AMD_SMALL_NTT_MODULUS
    TrlwePrimitives(eng, bk);
#endif
    RippleAdders(eng);
    CleanUp();
    std::printf("%s\n", g_failures ? "FAILED" : "ALL PASS");
    return g_failures ? 1 : 0;
#else
    static_assert(kParamSetIndex == 0, "the BASELINE numbers are the library's set 0");
    if (getenv("CUFHE_AMD_TEST_ONLY_STREAM_ORDERING")) {
        StreamOrdering(eng);
        CleanUp();
        std::printf("%s\n", g_failures ? "FAILED" : "ALL PASS");
        return g_failures ? 1 : 0;
    }
    if (getenv("CUFHE_AMD_TEST_QUICK")) {                   // the gate tests and the lvl2 keys only (the USE_TFHEPP build's run)
        Lvl2Gates(eng);
        CleanUp();
        std::printf("%s\n", g_failures ? "FAILED" : "ALL PASS");
        return g_failures ? 1 : 0;
    }
    Chained(eng);
    Intensive(eng);
    DeviceResident(eng);
    TrlwePrimitives(eng, bk);
    StreamOrdering(eng);
    RippleAdders(eng);
    if (!getenv("CUFHE_AMD_TEST_SKIP_AT_SIZE")) {           // the parts at size do not depend on the variant of the run (tests/test_gpu_parity.py)
        RefreshAtSize(eng);
        MixedAtSize(eng);
        Lvl2Gates(eng);
    }
    CleanUp();
    std::printf("%s\n", g_failures ? "FAILED" : "ALL PASS");
    return g_failures ? 1 : 0;
#endif
}
