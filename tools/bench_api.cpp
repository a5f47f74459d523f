// bench_api.cpp -- PCIe-inclusive rate of the reference-style per-gate API (not the headline metric).
// 4096 cufhe::Nand(out, a, b, st) calls on host-resident ciphertexts over 256 streams, then
// Synchronize(): what test/test_util.h:29-72 times in the reference ("Throughput: ms/gate").
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../include/cufhe_amd.hpp"

using namespace cufhe;
using P = TFHEpp::lvl0param;

int main(int argc, char** argv)
{
    const int kNumTests = argc > 1 ? atoi(argv[1]) : 4096, kNumStreams = 256;
    // further arguments: gpus=G (SetGPUNum(G): streams and gates round-robin the devices like the reference's
    // test_gate_gpu_multi.cc; with share_devices=1 the G logical devices share the visible GPUs) and library options as
    // key=value (cufhe_amd_set_option), e.g. sched_level_gates=4096
    // reps=R timed repetitions (the first is a warm-up), netlist=0 skips the adder netlist, identify=1 refuses to run when
    // the G logical devices are fewer than G distinct physical GPUs (unless share_devices=1 asked for exactly that)
    int gpus = 1, reps = 8, netlist = 1, identify = 0, pin = 0;
    long share = 0;
    for (int i = 2; i < argc; i++) {
        std::string kv(argv[i]);
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) continue;
        const std::string key = kv.substr(0, eq);
        if (key == "gpus") { gpus = atoi(kv.c_str() + eq + 1); continue; }
        if (key == "reps") { reps = std::max(2, atoi(kv.c_str() + eq + 1)); continue; }
        if (key == "netlist") { netlist = atoi(kv.c_str() + eq + 1); continue; }
        if (key == "identify") { identify = atoi(kv.c_str() + eq + 1); continue; }
        if (key == "pin") { pin = atoi(kv.c_str() + eq + 1); continue; }
        if (key == "share_devices") share = atol(kv.c_str() + eq + 1);
        CUFHE_AMD_CHECK(cufhe_amd_set_option(key.c_str(), atol(kv.c_str() + eq + 1)));
    }
    std::mt19937 eng(1);
    std::vector<uint32_t> bk((size_t)630 * 6 * 2 * 1024), ksk((size_t)1024 * 8 * 2 * 631);
    for (auto& v : bk) v = eng();
    for (auto& v : ksk) v = eng();
    if (identify && !share && cufhe_amd_device_count() < gpus) {
        std::fprintf(stderr, "bench_api: %d logical devices asked for, %d distinct GPU(s) visible; pass share_devices=1 to rehearse on this box\n",
                     gpus, cufhe_amd_device_count());
        return 3;
    }
    SetGPUNum(gpus);
    // which physical GPUs are these?  (cufhe_amd_device_identity: PCI function and UUID from the HIP runtime)
    std::vector<std::string> ident(gpus);
    std::set<std::string> distinct;
    for (int dev = 0; dev < gpus; dev++) {
        char buf[512];
        CUFHE_AMD_CHECK(cufhe_amd_device_identity(dev, buf, sizeof buf));
        ident[dev] = buf;
        const size_t u = ident[dev].find("uuid="), e = ident[dev].find(' ', u);
        distinct.insert(ident[dev].substr(u, e - u) == "uuid=?" ? ident[dev].substr(0, ident[dev].find(' ')) : ident[dev].substr(u, e - u));
    }
    if (identify && (int)distinct.size() < gpus && !share) {
        std::fprintf(stderr, "bench_api: %d logical devices on %zu distinct GPU(s); pass share_devices=1 to rehearse on this box\n", gpus, distinct.size());
        return 3;
    }
    // pin=1 (one GPU): the issuing thread (and the ciphertexts it allocates) on the CPUs close to the GPU, as an application placed with
    // numactl would be -- the identity string names them (local_cpus=0-63,128-191).  Default off: measured on a two-socket EPYC 9575F
    // box it makes no difference to the best repetition (38.2-38.4 ms pinned, 37.9-38.4 left to the OS; the launch worker and its
    // copy helpers are pinned by the library either way).
    int pinned_cpus = 0;
    if (gpus == 1 && pin) {
        const size_t at = ident[0].find("local_cpus=");
        cpu_set_t set;
        CPU_ZERO(&set);
        if (at != std::string::npos) {
            const char* q = ident[0].c_str() + at + 11;
            while (*q >= '0' && *q <= '9') {
                char* e;
                long lo = strtol(q, &e, 10), hi = lo;
                if (*e == '-') hi = strtol(e + 1, &e, 10);
                for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) { CPU_SET((int)c, &set); pinned_cpus++; }
                q = *e == ',' ? e + 1 : e;
            }
        }
        if (pinned_cpus == 0 || sched_setaffinity(0, sizeof set, &set) != 0) pinned_cpus = 0;
    }
    Initialize(bk.data(), bk.size(), ksk.data(), ksk.size());
    std::vector<Ctxt<P>> a(kNumTests), b(kNumTests), o(kNumTests);
    for (int i = 0; i < kNumTests; i++)
        for (auto* c : {&a[i], &b[i]})
            for (auto& w : c->tlwehost) w = eng();
    std::vector<Stream> st(kNumStreams);
    for (auto& s : st) s.Create();
    double best_tot = 1e30, best_enq = 0;
    cufhe_amd_sched_stats ss{};
    auto all_stats = [&](cufhe_amd_sched_stats& sum, int reset) {      // summed over the devices
        sum = cufhe_amd_sched_stats{};
        for (int dev = 0; dev < gpus; dev++) {
            cufhe_amd_sched_stats one{};
            CUFHE_AMD_CHECK(cufhe_amd_sched_get_stats(dev, &one, reset));
            sum.gates += one.gates; sum.levels += one.levels; sum.launch_sequences += one.launch_sequences;
            sum.record_ns += one.record_ns; sum.retire_ns += one.retire_ns; sum.launch_ns += one.launch_ns;
            sum.max_level_gates = std::max(sum.max_level_gates, one.max_level_gates);
            sum.two_lane_groups += one.two_lane_groups; sum.two_lane_launches += one.two_lane_launches;
        }
    };
    std::vector<cufhe_amd_sched_stats> per_dev(gpus);
    for (int rep = 0; rep < reps; rep++) {
        all_stats(ss, 1);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < kNumTests; i++) Nand(o[i], a[i], b[i], st[i % kNumStreams]);
        auto t1 = std::chrono::steady_clock::now();
        Synchronize();
        auto t2 = std::chrono::steady_clock::now();
        const double enq = std::chrono::duration<double, std::milli>(t1 - t0).count();
        const double tot = std::chrono::duration<double, std::milli>(t2 - t0).count();
        std::fprintf(stderr, "rep %d: %d Nand via per-gate API: enqueue %.2f ms, total %.2f ms, %.0f gates/s, %.4f ms/gate\n", rep,
                     kNumTests, enq, tot, kNumTests / (tot * 1e-3), tot / kNumTests);
        if (rep && tot < best_tot) { best_tot = tot; best_enq = enq; }
        for (int dev = 0; dev < gpus; dev++) CUFHE_AMD_CHECK(cufhe_amd_sched_get_stats(dev, &per_dev[dev], 0));
        all_stats(ss, 0);
    }
    // one JSON line (bench.py reads it): the PCIe-inclusive per-gate API rate and the host cost per gate --
    // on the issuing thread (record + deliver) and on the device's launch worker
    const double issue_us = (ss.record_ns + ss.retire_ns) * 1e-3 / kNumTests, worker_us = ss.launch_ns * 1e-3 / kNumTests;
    std::printf("{\"gates\": %d, \"devices\": %d, \"streams\": %d, \"total_ms\": %.3f, \"enqueue_ms\": %.3f, \"gates_per_s\": %.1f, "
                "\"host_issue_us_per_gate\": %.4f, \"host_worker_us_per_gate\": %.4f, \"host_issue_gates_per_s\": %.0f, "
                "\"launch_sequences\": %llu, \"distinct_gpus\": %zu, "
                // the three figures the reference's own harness prints (test/test_util.h:67-70): Total, Throughput = Total / gates,
                // Latency = Total / (gates per stream)
                "\"reference_style\": {\"total_ms\": %.3f, \"throughput_ms_per_gate\": %.5f, \"latency_ms_per_gate\": %.4f, \"gates_per_stream\": %d}, "
                "\"per_device\": [",
                kNumTests, gpus, kNumStreams, best_tot, best_enq, kNumTests / (best_tot * 1e-3), issue_us, worker_us, 1e6 / issue_us,
                (unsigned long long)ss.launch_sequences, distinct.size(),
                best_tot, best_tot / kNumTests, best_tot / std::max(1, kNumTests / kNumStreams), std::max(1, kNumTests / kNumStreams));
    for (int dev = 0; dev < gpus; dev++)
        std::printf("%s{\"device\": %d, \"gpu\": \"%s\", \"gates\": %llu, \"launch_sequences\": %llu, \"worker_launch_ms\": %.3f, "
                    "\"worker_pinned_cpus\": %llu}", dev ? ", " : "", dev, ident[dev].c_str(), (unsigned long long)per_dev[dev].gates,
                    (unsigned long long)per_dev[dev].launch_sequences, per_dev[dev].launch_ns * 1e-6, (unsigned long long)per_dev[dev].worker_cpus);
    std::printf("], \"issuing_thread_pinned_cpus\": %d", pinned_cpus);
    {
        // one more repetition with HIP timing events on (cufhe_amd_profile_enable) for the timeline of the run: where the
        // milliseconds between the first Nand() and the return of Synchronize() go, flush by flush (device 0)
        for (int dev = 0; dev < gpus; dev++) CUFHE_AMD_CHECK(cufhe_amd_profile_enable(dev, 1));
        std::vector<cufhe_amd_group_trace> tr(64);
        (void)cufhe_amd_sched_get_trace(0, tr.data(), 64, 1);
        const int64_t t0 = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        for (int i = 0; i < kNumTests; i++) Nand(o[i], a[i], b[i], st[i % kNumStreams]);
        const int64_t t1 = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        Synchronize();
        const int64_t t2 = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        for (int dev = 0; dev < gpus; dev++) CUFHE_AMD_CHECK(cufhe_amd_profile_enable(dev, 0));
        const int n = cufhe_amd_sched_get_trace(0, tr.data(), 64, 1);
        auto ms = [&](int64_t t) { return (t - t0) * 1e-6; };
        std::printf(", \"timeline\": {\"what\": \"one repetition with timing events on; host times in ms from the first Nand() call, device 0\", "
                    "\"enqueue_end_ms\": %.3f, \"synchronize_return_ms\": %.3f, \"flushes\": [", ms(t1), ms(t2));
        for (int i = 0; i < n; i++)
            std::printf("%s{\"gates\": %u, \"in_kib\": %llu, \"out_kib\": %llu, \"queued_ms\": %.3f, \"worker_begin_ms\": %.3f, \"inputs_gathered_ms\": %.3f, "
                        "\"submitted_ms\": %.3f, \"completion_seen_ms\": %.3f, \"delivered_ms\": %.3f, \"dev_h2d_scatter_ms\": %.3f, \"dev_gates_ms\": %.3f, "
                        "\"dev_gather_d2h_ms\": %.3f}", i ? ", " : "", tr[i].gates, (unsigned long long)(tr[i].in_bytes >> 10),
                        (unsigned long long)(tr[i].out_bytes >> 10), ms(tr[i].t_queued), ms(tr[i].t_launch_begin), ms(tr[i].t_gather_end),
                        ms(tr[i].t_submit_end), ms(tr[i].t_done_seen), ms(tr[i].t_delivered), tr[i].dev_h2d_ms, tr[i].dev_body_ms, tr[i].dev_d2h_ms);
        std::printf("]}");
    }
    std::printf("}\n");
    std::fflush(stdout);
    // A depth-first netlist through the same API: 256 independent 16-bit ripple-carry adders, issued ADDER BY ADDER
    // (every gate depends on the previous ones of its adder; test/test_api_gpu.cu:140-159 is the pattern in small).
    // The scheduler cuts the 20 480 recorded gates into dependence levels across the adders; once with the defaults
    // ("sched_rename" 1: outputs take fresh device buffers, only the carry chain is left, the values return to the
    // ciphertexts' own buffers in the flush of Synchronize) and once as the reference's buffers dictate ("sched_rename" 0:
    // the temporaries t1, t2 re-used bit after bit order the program).
    // variants: the defaults (renaming, per-gate scheduling on two lanes where the cost model picks it), the same level by level
    // ("sched_two_lane" 0), and as the reference's buffers dictate ("sched_rename" 0)
    for (int variant = 0; variant < 3 && netlist; variant++) {
        const int rename = variant < 2 ? 1 : 0, two_lane = variant == 0 ? 1 : 0;
        CUFHE_AMD_CHECK(cufhe_amd_set_option("sched_two_lane", two_lane));
        CUFHE_AMD_CHECK(cufhe_amd_set_option("sched_rename", rename));
        const int kAdders = 256, kBits = 16;
        std::vector<Ctxt<P>> x(kAdders * kBits), y(kAdders * kBits), sum(kAdders * kBits), carry(kAdders), t1(kAdders), t2(kAdders);
        for (auto* v : {&x, &y})
            for (auto& c : *v)
                for (auto& w : c.tlwehost) w = eng();
        for (auto& c : carry)
            for (auto& w : c.tlwehost) w = eng();
        double best = 1e30;
        cufhe_amd_sched_stats ns{};
        for (int rep = 0; rep < 3; rep++) {
            all_stats(ns, 1);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < kAdders; i++) {
                Stream s = st[i % kNumStreams];
                for (int k = 0; k < kBits; k++) {
                    Ctxt<P>&X = x[i * kBits + k], &Y = y[i * kBits + k], &S = sum[i * kBits + k], &C = carry[i];
                    Xor(t1[i], X, Y, s);
                    Xor(S, t1[i], C, s);
                    And(t2[i], t1[i], C, s);
                    And(t1[i], X, Y, s);
                    Or(C, t1[i], t2[i], s);
                }
            }
            Synchronize();
            best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            all_stats(ns, 0);
        }
        std::printf("{\"netlist\": \"256 x 16-bit ripple-carry adders, issued depth-first\", \"sched_rename\": %d, \"sched_two_lane\": %d, \"gates\": %llu, "
                    "\"total_ms\": %.2f, \"gates_per_s\": %.0f, \"dependence_levels\": %llu, \"launch_sequences\": %llu, "
                    "\"max_level_gates\": %llu, \"two_lane_flushes\": %llu, \"two_lane_launches\": %llu}\n",
                    rename, two_lane, (unsigned long long)ns.gates, best, ns.gates / (best * 1e-3), (unsigned long long)ns.levels,
                    (unsigned long long)ns.launch_sequences, (unsigned long long)ns.max_level_gates,
                    (unsigned long long)ns.two_lane_groups, (unsigned long long)ns.two_lane_launches);
    }
    CUFHE_AMD_CHECK(cufhe_amd_set_option("sched_rename", 1));
    CUFHE_AMD_CHECK(cufhe_amd_set_option("sched_two_lane", 1));
    // What a smarter level assignment could reach at best, emulated by the program itself: the same adders in single-assignment
    // form (no buffer is written twice, so only data dependences order it), the sum bits -- the only gates nothing else reads --
    // withheld until the carry chains are launched and issued as ONE level at the end.  If this is not faster than the
    // scheduler's own levels, moving gates between levels cannot be either (launch shapes).
    if (netlist) {
        const int kAdders = 256, kBits = 16;
        const size_t nb = (size_t)kAdders * kBits;
        std::vector<Ctxt<P>> x(nb), y(nb), sum(nb), prop(nb), gen(nb), t2(nb), carry((size_t)kAdders * (kBits + 1));
        for (auto* v : {&x, &y})
            for (auto& c : *v)
                for (auto& w : c.tlwehost) w = eng();
        for (int i = 0; i < kAdders; i++)
            for (auto& w : carry[(size_t)i * (kBits + 1)].tlwehost) w = eng();
        double best = 1e30;
        cufhe_amd_sched_stats ns{};
        for (int rep = 0; rep < 3; rep++) {
            all_stats(ns, 1);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < kAdders; i++) {
                Stream s = st[i % kNumStreams];
                for (int k = 0; k < kBits; k++) {
                    const size_t b = (size_t)i * kBits + k, c = (size_t)i * (kBits + 1) + k;
                    Xor(prop[b], x[b], y[b], s);
                    And(gen[b], x[b], y[b], s);
                    And(t2[b], prop[b], carry[c], s);
                    Or(carry[c + 1], gen[b], t2[b], s);
                }
            }
            for (int dev = 0; dev < gpus; dev++) CUFHE_AMD_CHECK(cufhe_amd_flush(dev));      // the chains are on their way
            for (int i = 0; i < kAdders; i++)
                for (int k = 0; k < kBits; k++) {
                    const size_t b = (size_t)i * kBits + k, c = (size_t)i * (kBits + 1) + k;
                    Xor(sum[b], prop[b], carry[c], st[i % kNumStreams]);
                }
            Synchronize();
            best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            all_stats(ns, 0);
        }
        std::printf("{\"netlist\": \"the same adders in single-assignment form, sum bits issued as one level behind the carry chains (the bound of any level re-assignment)\", "
                    "\"gates\": %llu, \"total_ms\": %.2f, \"gates_per_s\": %.0f, \"dependence_levels\": %llu, \"launch_sequences\": %llu, \"max_level_gates\": %llu}\n",
                    (unsigned long long)ns.gates, best, ns.gates / (best * 1e-3), (unsigned long long)ns.levels,
                    (unsigned long long)ns.launch_sequences, (unsigned long long)ns.max_level_gates);
    }
    for (auto& s : st) s.Destroy();
    CleanUp();
    return 0;
}
