#!/usr/bin/env python3
"""bench.py -- NAND gate-bootstraps/s of the HIP gate path (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: 4096 independent NAND gates
(BASELINE.json configs[1]) per GPU, inputs and keys already resident in HBM, one
cufhe_amd_gate_batch call = one blind-rotate launch + one key-switch launch.  With N > 1
ranks (one per GPU) every rank runs its own 4096 gates against its own BK/KSK replica: weak
scaling, no data-path collective.  `python bench.py --gpus N` run plainly starts
the N ranks itself (before torch or the HIP library are imported); under
`torch.distributed.run` it is one of the ranks.  `--workload mixed --total-gates 32768` is
BASELINE configs[2] instead: the 32 768 mixed gates of SURVEY.md 8(d) config 3 split
contiguously over the ranks (strong scaling); a plain `--gpus N` run with N > 1 times that
workload as well, in the same processes, after the NAND timing (extra_workloads.mixed_32768_strong).

The line starts with the contract fields and ENDS with a compact `summary` (the driver's record keeps the
tail of stdout): value, value_pcie_inclusive and their ratio, single-gate latency, one rate per extra
workload, the dependent netlist, how the word checks went.  What the fields mean and how each is
measured is prose that does not change from run to run: profiles/bench_line_notes.md.

Prints ONE JSON line (rank 0):
  value / ms_per_step      the timed region: K steps, no profiling hooks inside
  value_pcie_inclusive     the same 4096 gates through the reference-style per-gate API on host-resident
                           ciphertexts, enqueue -> Synchronize; details in
                           api_pcie_inclusive (+ a depth-first adder netlist, both renaming modes)
  ms_per_gate_latency_single_gate   one gate alone on the idle device (the metric's second half)
  roofline                 the dominant kernel (blind rotate) priced with the BK-sweep accounting of
                           SURVEY.md 8(d) (61 931 520 algorithmic bytes per rotation) over its launch
                           time, measured with HIP events on the launch stream in a SEPARATE pass;
                           `bound` names the real limiter (FP64 vector issue) and `valu` prices it:
                           instruction count and HBM bytes per rotation from the committed rocprofv3
                           PMC passes of this command (valid only while every file with device code
                           hashes to what was profiled), launch time and shader clock from THIS run
  extra_workloads          mux (configs[3]), mixed (configs[2] op mix), nand_lvl2 (configs[4]), a
                           512-gate launch (paired low-latency kernel), the key switch: rate, roofline
                           fraction, `valu` block and a word-for-word oracle check each; param_sets: the
                           other compiled parameter sets through the generic kernels
  cpu_baseline             an optimised CPU implementation of the same gate (oracle/cpu_fast.c: same
                           exact FP64 field, AVX-512/AVX2, OpenMP over gates) timed on this box's host
                           cores on a bounded sample; the CPU oracle checks both its words and the GPU's
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# one process per GPU: rank r drives physical device r of the node (device_base option below).
# cufhe_amd/dist.py is pure Python; it is loaded by path so that the package (and with it the
# HIP library) is imported only after torch.
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("cufhe_amd_dist", os.path.join(ROOT, "cufhe_amd", "dist.py"))
distutil = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(distutil)
RANK, LOCAL_RANK, WORLD = distutil.rank_env()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")



def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gates", type=int, default=4096, help="gates per GPU per step")
    ap.add_argument("--total-gates", type=int, default=0,
                    help="strong scaling: this many gates per step split contiguously over the ranks (configs[2]: 32768)")
    ap.add_argument("--mode", choices=["ranks", "api"], default="ranks",
                    help="ranks = one process per GPU (the driver's form; default).  api = ONE process drives --gpus G devices through "
                         "SetGPUNum(G) and round-robin Streams, the reference's own multi-GPU shape (include/cufhe_gpu.cuh:154-165, "
                         "test/test_gate_gpu_multi.cc:36-93): per-gate API on host ciphertexts, PCIe-inclusive")
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="let ranks (or logical devices) share a physical GPU: a rehearsal, labelled shared_gpu in the line with "
                         "n_gpus = the distinct GPUs; without it such a run exits non-zero")
    ap.add_argument("--no-api", action="store_true", help="skip the per-gate API subprocess (profiling passes: one launch shape per kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_workloads / api_pcie_inclusive / latency")
    ap.add_argument("--workload", choices=["nand", "mux", "mixed", "nand_lvl2"], default="nand",
                    help="nand = BASELINE configs[1] (the metric's config); mux = configs[3]; mixed = configs[2] op mix; "
                         "nand_lvl2 = configs[4] (N = 2048 ring, 64-bit torus)")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` run plainly (N > 1, no WORLD_SIZE in the environment): this process
    becomes the launcher -- it has not imported torch or the HIP library and never touches the GPU --
    starts N fresh rank processes of this same command, relays rank 0's JSON line and exits with
    the first failing rank's code (cufhe_amd/dist.py: spawn_ranks)."""
    timeout = float(os.environ.get("CUFHE_AMD_BENCH_LAUNCH_TIMEOUT", "1500"))
    rc, out = distutil.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout=timeout)
    sys.stdout.write(out)
    sys.stdout.flush()
    if rc == 0 and not any(l.startswith("{") for l in out.splitlines()):
        sys.stderr.write("bench.py launcher: rank 0 printed no JSON line\n")
        rc = 1
    sys.exit(rc)


def api_mode(args):
    """`--mode api --gpus G`: ONE process, SetGPUNum(G), Streams round-robin over the devices -- the reference's
    multi-GPU shape (include/cufhe_gpu.cuh:154-165; test/test_gate_gpu_multi.cc:36-93).  The host side is C++ over the C
    ABI (tools/bench_api.cpp, the cuFHE names of include/cufhe_amd.hpp); this process only starts it and relays one line.
    A step = gates * G cufhe::Nand calls on host-resident ciphertexts over 256 streams, then Synchronize(): H2D and D2H of
    every ciphertext are inside the timed region (test/test_util.h:29-72), so `value` here is the PCIe-inclusive rate."""
    exe = os.path.join(ROOT, "tools", "bench_api")
    total = args.gates * args.gpus
    cmd = [exe, str(total), "gpus=%d" % args.gpus, "reps=%d" % (args.steps + args.warmup), "netlist=0", "identify=1"]
    if args.allow_shared_gpu:
        cmd.append("share_devices=1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get("CUFHE_AMD_BENCH_LAUNCH_TIMEOUT", "1500")))
    sys.stderr.write(p.stderr[-4000:])
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        sys.stderr.write("bench.py --mode api: tools/bench_api exited with %d\n" % p.returncode)
        sys.exit(p.returncode or 1)
    r = lines[0]
    res = {
        "metric": "nand_gate_bootstraps_per_sec", "value": r["gates_per_s"], "unit": "gate-bootstraps/s",
        "n_gpus": r["distinct_gpus"], "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["total_ms"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.gates} independent NAND gates per GPU per step (BASELINE configs[1] per device), one process: "
                               f"SetGPUNum({args.gpus}), 256 Streams round-robin over the devices, per-gate API on HOST ciphertexts "
                               "(PCIe-inclusive: enqueue -> Synchronize), TFHE n=630 N=1024 k=1 l=3 Bgbit=6 t=8 basebit=2",
                   "mode": "api", "gates_per_gpu": args.gates, "logical_devices": args.gpus,
                   "sharding": "stream i -> device i mod G (include/cufhe_gpu.cuh:154-159), per-GPU BK/KSK replica, no collective"},
        "per_device": r.get("per_device"), "distinct_gpus": r["distinct_gpus"],
        "value_is": "PCIe-inclusive (best of the timed repetitions); the inputs-resident metric of the driver line is --mode ranks",
        "api": r,
    }
    if r["distinct_gpus"] < args.gpus:
        res["shared_gpu"] = True
    print(json.dumps(res), flush=True)
    sys.exit(0)


if __name__ == "__main__":
    ARGS = parse_args()
    if ARGS.mode == "api":
        api_mode(ARGS)               # does not return; this process never touches the GPU
    if "WORLD_SIZE" not in os.environ and ARGS.gpus > 1:
        self_launch(ARGS)            # does not return

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: its bundled HIP runtime is the one the process uses)
import torch.distributed as dist  # noqa: E402

BK_BYTES_PER_ROTATION = 61931520          # n (k+1)^2 l N 8, SURVEY.md 8(d)
BK2_BYTES_PER_ROTATION = 630 * 4 * 4 * 2048 * 8   # the same accounting at N = 2048, l = 4
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8 TB/s spec
SIMDS = 1024                              # 256 CUs x 4


FP64_ISSUE_CYCLES_SPEC = 4.0              # data sheet: 64 lanes over a 16-lane FP64 pipe (78.6 TFLOP/s)
FP64_ISSUE_CYCLES_MEASURED = 4.34         # tools/ubench/ubench_ilp.hip on MI355X: two waves per SIMD, v_fma_f64 back to back
# device code AND the host files that decide launch shapes (units per workgroup are part of what the PMC facts assume)
DEVICE_SOURCES = ("fpfield.h", "ntt_r4.h", "ntt_wave.h", "ntt_wave512.h", "kernels_common.hip.h", "kernels.hip.h", "kernels_ll.hip.h",
                  "kernels_lvl2.hip.h", "kernels_lvl2q.hip.h", "kernels_ks2.hip.h", "kernels_ps.hip.h", "capi.hip", "lvl2.inc.h", "paramsets.inc.h")


def source_hash():
    """sha256 over every file with device code: recorded PMC facts are only valid for the exact kernels."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cufhe_amd", "csrc")
    for f in DEVICE_SOURCES:
        if not os.path.exists(os.path.join(d, f)):
            continue
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


_FACTS = None


def kernel_facts(kernel):
    """Counters of one kernel from the committed rocprofv3 PMC passes (profiles/kernel_facts.json, written by
    tools/kernel_facts.py from separate --pmc runs of this command): instruction counts and HBM bytes per rotation, which
    do not depend on the run; clocks and times always come from THIS run.  None when the device code changed since."""
    global _FACTS
    if _FACTS is None:
        try:
            _FACTS = json.load(open(os.path.join(ROOT, "profiles", "kernel_facts.json")))
        except Exception:
            _FACTS = {}
    if not _FACTS:
        return None, "no recorded PMC passes"
    if _FACTS.get("source_sha256") != source_hash():
        return None, "stale: cufhe_amd/csrc changed since the PMC passes were recorded (%s)" % _FACTS.get("recorded", "?")
    k = _FACTS.get("kernels", {}).get(kernel)
    if not k:
        return None, "no PMC pass for this kernel"
    return k, _FACTS.get("recorded", "")


def replayed_from(passes):
    """names of the committed rocprofv3 CSVs a replayed figure comes from (profiles/kernel_facts.json lists them)"""
    src = (_FACTS or {}).get("sources") or []
    pick = [f for f in src if any(("_pmc_%s_" % p) in f for p in passes)]
    return ["profiles/" + f for f in (pick or src)] or "profiles/kernel_facts.json"


def valu_block(kernel, rotations, launch_ms, clock_hz):
    """What actually bounds these kernels: FP64 VALU issue.  Instruction count per rotation from the PMC pass (a property
    of the code), launch time and shader clock from this run."""
    facts, note = kernel_facts(kernel)
    if not facts or launch_ms <= 0 or not clock_hz:
        return {"pmc_source": note}
    insts = facts["valu_insts_per_rotation"] * rotations
    slots = SIMDS * clock_hz * launch_ms * 1e-3          # SIMD-cycles available in the launch
    return {
        "insts_per_rotation": facts["valu_insts_per_rotation"],
        "insts_per_step_per_wave": facts.get("valu_insts_per_step_per_wave"),
        "frac_of_fp64_issue": FP64_ISSUE_CYCLES_SPEC * insts / slots,
        "frac_of_measured_issue_floor": FP64_ISSUE_CYCLES_MEASURED * insts / slots,
        "clock_hz_this_run": clock_hz,
        "clock_hz_under_profiler": facts.get("clock_hz_under_profiler"),
        "pipe_busy_under_profiler": facts.get("valu_pipe_busy"),
        "lds_bank_conflict_frac": facts.get("lds_bank_conflict_frac"),
        "waves_per_simd": facts.get("waves_per_simd"),
        "pmc_source": note,
        "replayed_from": replayed_from(("sq", "lds", "mix")),
    }


def oracle_check(ol, L, ek, ops, in0, in1, in2, gpu_out, idx):
    """GPU words of the sampled gates == CPU oracle words (the checker, not the thing measured)."""
    idx = np.asarray(idx)
    ops_arr = np.ascontiguousarray(ops[idx] if isinstance(ops, np.ndarray) else np.full(idx.size, ops), np.int32)
    out = np.zeros(idx.size * (ol.n + 1), np.uint32)
    a = np.ascontiguousarray(in0[idx]).ravel()
    b = np.ascontiguousarray(in1[idx])
    c = np.ascontiguousarray(in2[idx])
    L.orc_gate_batch(ek, ops_arr, 1, 0, idx.size, out, a, b.ctypes.data, c.ctypes.data, min(L.orc_max_threads(), 32))
    return bool(np.array_equal(out.reshape(idx.size, -1), gpu_out[idx]))


def cgroup_cpu_quota():
    """CPUs this container may use according to its cgroup (v2 cpu.max, v1 cfs quota), or None when unlimited / unreadable."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_baseline(ol, L, bk, ksk, in0, in1, gpu_out, oracle_ek, target_seconds=15.0):
    """Time the optimised CPU gate (oracle/cpu_fast.c) on a bounded sample of the same workload: on every core this process may
    use (all visible cores unless the cgroup quota says fewer) and, measured separately, on ONE thread."""
    fek = L.fast_evalkey_create(bk, ksk)
    visible = L.orc_max_threads()
    quota = cgroup_cpu_quota()
    nand = np.array([0], np.int32)
    words = ol.n + 1

    def run(count, threads):
        out = np.zeros(count * words, np.uint32)
        a = np.ascontiguousarray(in0[:count]).ravel()
        b = np.ascontiguousarray(in1[:count]).ravel()
        t = time.perf_counter()
        L.fast_gate_batch(fek, nand, 0, count, out, a, b, threads)
        return time.perf_counter() - t, out.reshape(count, words)

    threads = visible if quota is None else max(1, min(visible, int(quota + 0.5)))
    # one thread, measured (not derived from the parallel run): 12 gates after one to warm the tables
    run(1, 1)
    dt1, _ = run(12, 1)
    single_ms = 1e3 * dt1 / 12
    # one round per candidate thread count (16 gates per thread: one block), for context: the share of the box a container really
    # gets can be below what it is shown, and then fewer threads are faster
    calib = []
    for cand in sorted({min(16, visible), min(32, visible), min(64, visible), visible, threads}):
        dt, _ = run(16 * cand, cand)
        calib.append({"threads": int(cand), "gates_per_s": round(16 * cand / dt, 1)})
    rate = next(c["gates_per_s"] for c in calib if c["threads"] == threads)
    # about target_seconds of CPU work: whole passes over the batch (the same gates the GPU ran)
    count = in0.shape[0]
    passes = max(1, int(round(target_seconds * rate / count)))
    dt = 0.0
    for _ in range(passes):
        d, out = run(count, threads)
        dt += d
    L.fast_evalkey_destroy(fek)
    # the checker: the restatement of the reference's algorithm on a handful of the same gates
    idx = np.arange(0, count, max(1, count // 32))[:32]
    want = np.zeros(idx.size * words, np.uint32)
    a_s, b_s = np.ascontiguousarray(in0[idx]).ravel(), np.ascontiguousarray(in1[idx]).ravel()   # kept alive across the call
    L.orc_gate_batch(oracle_ek, nand, 0, 0, idx.size, want, a_s, b_s.ctypes.data, None, min(visible, 32))
    want = want.reshape(idx.size, words)
    count_total = count * passes
    return {
        "value": count_total / dt, "unit": "gate-bootstraps/s", "cores": int(threads), "kind": "port",
        "sample": f"{passes} pass(es) over the batch's {count} NAND gates on {threads} threads, {dt:.1f} s; 12 gates on one thread, {dt1:.2f} s",
        "single_thread_ms_per_gate": single_ms,
        "visible_cores": int(visible), "cgroup_cpu_quota": quota,
        "implementation": "oracle/cpu_fast.c: exact FP64-field NTT, constant-geometry radix-2, "
                          + {4: "AVX-512", 3: "AVX2+FMA", 0: "scalar"}[int(L.fast_isa_level())] + ", OpenMP over gates",
        "ms_per_gate_per_core_in_the_parallel_run": 1e3 * dt * threads / count_total,
        "thread_count_rounds": calib,
        "cpu_words_match_oracle": bool(np.array_equal(out[idx], want)),
        "gpu_words_match_cpu": bool(np.array_equal(out, gpu_out[:count])),        # all 4096 gates
        "gpu_words_match_oracle": bool(np.array_equal(gpu_out[idx], want)),
        "reference_readme_context": "TFHE library on CPU: 10 ms per gate; cuFHE: 13 ms per gate per A100 SM (README.md:29-31)",
    }


def ordered_line(res):
    """Key order of the one JSON line.  The contract fields come first, then the roofline and cpu_baseline objects cut to their numbers,
    the detail blocks, and -- as the LAST key, because the driver's record keeps the TAIL of stdout -- a compact `summary`: the
    PCIe-inclusive rate and its ratio to `value`, the single-gate latency, one figure per extra workload, the dependent netlist with
    its launch-sequence count, and how the word checks went (run / failed / unchecked / workloads that raised)."""
    ex = res.get("extra_workloads") or {}
    api_blk = res.get("api_pcie_inclusive") or {}

    def val(d, k="value"):
        return d.get(k) if isinstance(d, dict) else None
    netl = api_blk.get("depth_first_netlist") if isinstance(api_blk, dict) else None
    summary = {
        "value": res.get("value"), "value_pcie_inclusive": res.get("value_pcie_inclusive"),
        "pcie_inclusive_over_value": (res["value_pcie_inclusive"] / res["value"]) if res.get("value_pcie_inclusive") and res.get("value") else None,
        "single_gate_ms": res.get("ms_per_gate_latency_single_gate"),
        "roofline_frac": val(res.get("roofline"), "frac"),
        "mux_gates_per_s": val(ex.get("mux")), "mixed_gates_per_s": val(ex.get("mixed")),
        "nand_lvl2_per_s": val(ex.get("nand_lvl2")), "nand_level1_per_s": val(ex.get("nand_level1")),
        "nand_512_ms": val(ex.get("nand_512"), "ms_per_step"),
        "mixed_32768_strong_gates_per_s": val(ex.get("mixed_32768_strong")),
        "adder_netlist_gates_per_s": val(netl, "gates_per_s"),
        "adder_netlist_over_value": (netl["gates_per_s"] / res["value"]) if isinstance(netl, dict) and netl.get("gates_per_s") and res.get("value") else None,
        "adder_netlist_launch_sequences": val(netl, "launch_sequences"),
        "adder_netlist_two_lane_launches": val(netl, "two_lane_launches"),
        "adder_netlist_level_by_level_gates_per_s": val(api_blk.get("depth_first_netlist_level_by_level"), "gates_per_s"),
        "adder_netlist_without_renaming_gates_per_s": val(api_blk.get("depth_first_netlist_without_renaming"), "gates_per_s"),
        "api_reference_style_latency_ms_per_gate": val(api_blk.get("reference_style"), "latency_ms_per_gate"),
        "api_single_nand_ms": val(api_blk, "single_nand_call_to_synchronize_ms"),
        "api_chain16_ms_per_gate": val(api_blk, "chain_of_16_dependent_nand_ms_per_gate"),
        "param_sets_per_s": {k: val(v) for k, v in (ex.get("param_sets") or {}).items()} or None,
    }
    # Word checks: every "*_match_oracle" / "*_match_cpu" field anywhere in the record.  True counts as passed; False as failed; None
    # (the check could not be made, e.g. --no-cpu-baseline) as unchecked; a workload recorded as {"error": ..} as an error.  The flag
    # is true only when something was checked and nothing failed, was left unchecked where a check was due, or raised.
    counts = {"run": 0, "failed": 0, "unchecked": 0, "errors": 0}

    def walk(o):
        if isinstance(o, dict):
            if "error" in o and isinstance(o["error"], str):
                counts["errors"] += 1
            for k, v in o.items():
                if k.endswith("_match_oracle") or k.endswith("_match_cpu"):
                    if v is None:
                        counts["unchecked"] += 1
                    else:
                        counts["run"] += 1
                        counts["failed"] += 0 if v else 1
                else:
                    walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
    walk(res)
    summary["word_checks"] = counts
    summary["all_word_checks_pass"] = bool(counts["run"]) and not (counts["failed"] or counts["unchecked"] or counts["errors"])
    summary = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in summary.items() if v is not None}
    head_keys = ("metric", "value", "unit", "n_gpus", "ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "value_pcie_inclusive", "ms_per_gate_latency_single_gate", "ms_per_gate_throughput")
    out = {k: res[k] for k in head_keys if k in res}
    rf = res.get("roofline")
    if rf:
        first = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "frac_valu_fp64", "launch_ms", "rotations_per_launch",
                 "algorithmic_bytes_per_rotation")
        out["roofline"] = {k: rf[k] for k in first if k in rf}
    cb = res.get("cpu_baseline")
    if cb:
        first = ("value", "unit", "cores", "kind", "sample", "single_thread_ms_per_gate", "visible_cores", "cgroup_cpu_quota")
        out["cpu_baseline"] = {k: cb[k] for k in first if k in cb}
    if "config" in res:
        out["config"] = res["config"]
    if rf:
        out["roofline_detail"] = {k: v for k, v in rf.items() if k not in out["roofline"]}
    if cb:
        out["cpu_baseline_detail"] = {k: v for k, v in cb.items() if k not in out["cpu_baseline"]}
    for k, v in res.items():
        if k not in out and k != "summary":
            out[k] = v
    out["summary"] = summary          # last: what the tail of the line shows
    return out


def main():
    args = ARGS
    # stdout carries ONE line, the JSON: whatever else writes to file descriptor 1 below (gloo's "[Gloo] Rank 0 is connected ..."
    # banner, runtime notices) goes to stderr
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.gpus != WORLD:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={WORLD}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                 "or plainly (no WORLD_SIZE in the environment) and bench.py starts the ranks itself")

    if WORLD > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=RANK, world_size=WORLD)
    DEV = distutil.device_for_rank(LOCAL_RANK, torch.cuda.device_count()) if WORLD > 1 else 0
    torch.cuda.set_device(DEV)
    torch.cuda.init()

    import cufhe_amd as eng                  # fails loudly if the HIP library is missing
    eng.api.set_option("device_base", DEV)   # logical device 0 of this process = this rank's GPU
    api = eng.api
    # Which PHYSICAL GPU is this rank on?  device_for_rank wraps when fewer devices are visible than ranks, and a line that says
    # n_gpus = N must have been measured on N GPUs: every rank reports the PCI function / UUID of its device (HIP runtime, through
    # the C ABI), all ranks see the gathered list and all take the same decision before any work is done.
    identity = api.device_identity(0)
    identities = distutil.gather_objects(identity, dist)
    try:
        distinct_gpus, shared_gpu = distutil.check_distinct_gpus(identities, args.allow_shared_gpu)
    except distutil.SharedGpuError as e:
        if RANK == 0:
            sys.stderr.write("bench.py: %s\n" % e)
        if WORLD > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3)

    class ol:                                 # sizes come from the library, not from the oracle
        n, N = int(eng.PARAMS.n), int(eng.PARAMS.N)
        BK_WORDS, KSK_WORDS = int(eng.PARAMS.bk_words), int(eng.PARAMS.ksk_words)

    strong = args.total_gates > 0
    if strong:
        lo, hi = distutil.shard(args.total_gates, RANK, WORLD)     # gate i -> rank floor(i / ceil(total / W))
        count, first_gate = hi - lo, lo
    else:
        count, first_gate = args.gates, 0
    # synthetic keys and ciphertexts: uniform torus words (the path's work is data-independent).  Keys are
    # the same on every rank (per-GPU replicas of ONE key set, as the reference uploads them); inputs differ.
    krng = np.random.default_rng(4242)
    rng = np.random.default_rng(42 + (0 if strong else RANK))
    lvl2 = args.workload == "nand_lvl2"
    bk = krng.integers(0, 2**32, size=ol.BK_WORDS, dtype=np.uint64).astype(np.uint32)
    ksk = krng.integers(0, 2**32, size=ol.KSK_WORDS, dtype=np.uint64).astype(np.uint32)
    total = args.total_gates if strong else count
    ins_all = [rng.integers(0, 2**32, size=(total, ol.n + 1), dtype=np.uint64).astype(np.uint32) for _ in range(3)]
    in0, in1, in2 = (x[first_gate:first_gate + count] for x in ins_all)

    eng.SetGPUNum(1)
    eng.Initialize(bk, ksk)
    p2 = bk2 = ksk2 = None

    def init_lvl2():
        nonlocal p2, bk2, ksk2
        p2 = api.lvl2_params()
        r2 = np.random.default_rng(4343)
        bk2 = r2.integers(0, 2**64, size=int(p2.bk_words), dtype=np.uint64)
        ksk2 = r2.integers(0, 2**32, size=int(p2.ksk_words), dtype=np.uint64).astype(np.uint32)
        api.lvl2_initialize(bk2, ksk2)

    if lvl2:
        init_lvl2()
    d0, d1, d2 = (api.DeviceBuffer(max(x.size, 1)).upload(x) for x in (in0, in1, in2))
    dout = api.DeviceBuffer(max(count, 1) * (ol.n + 1))
    st = eng.Stream(0)
    st.Create()
    mixed_ops = np.array([[api.AND, api.OR, api.XOR, api.NAND][(first_gate + g) % 4] for g in range(count)], np.int32)

    def run_step(workload, n=count):
        if n == 0:
            return
        if workload == "nand_lvl2":
            api.lvl2_gate_batch(api.NAND, dout, d0, d1, d2, count=n, device=0, stream=st.st())
        else:
            ops = {"nand": api.NAND, "mux": api.MUX, "mixed": mixed_ops}[workload]
            api.gate_batch(ops, 0, dout, d0, d1, d2, count=n, device=0, stream=st.st())

    def barrier():
        if WORLD > 1:
            dist.barrier()
        eng.Synchronize()
        torch.cuda.synchronize()

    own = {}

    def timed(workload, steps, warmup):
        """K steps bracketed by barrier + synchronize on both sides; own["s"] = this rank's own time (its GPU idle again,
        before it waits for the other ranks)"""
        for _ in range(warmup):
            run_step(workload)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step(workload)
        eng.Synchronize()
        torch.cuda.synchronize()
        own["s"] = time.perf_counter() - t0
        barrier()
        return time.perf_counter() - t0

    def kernel_times(workload, steps=2):
        """launch durations from HIP events on the launch stream, in a pass of their own"""
        eng.Synchronize()
        api.profile_get(reset=True)
        api.profile_enable(True)
        for _ in range(steps):
            run_step(workload)
        eng.Synchronize()
        prof = api.profile_get(reset=True)
        api.profile_enable(False)
        return (prof.blind_rotate_ms / max(prof.blind_rotate_launches, 1), prof.keyswitch_ms / max(prof.keyswitch_launches, 1))

    clocks = {}

    def clock_during(key, launch, launch_ms=0.0):
        """The shader clock the chip holds UNDER THIS WORKLOAD: the probe kernel (cufhe_amd_probe_clock: shader cycles over the constant
        100 MHz counter, ~1 ms of dependent FMAs, no LDS, a handful of registers) is launched on the null stream while the workload's launch
        runs on its own non-blocking stream, so its waves sit in the register space the big kernel leaves and count the clock that kernel
        really gets.  A probe on an otherwise idle chip reads 2.43 GHz; under blind_rotate_kernel the chip holds ~2.35, under
        blind_rotate_lvl2q_kernel ~2.15-2.18 (power): pricing `valu` with the idle figure understates how busy the pipe is."""
        if key not in clocks:
            eng.Synchronize()
            launch()
            time.sleep(0.4e-3 * launch_ms)          # sample the middle of the launch: its first milliseconds are a power-management transient (1.8 GHz)
            clocks[key] = api.probe_clock()
            eng.Synchronize()
        return clocks[key]

    def roofline(workload, br_ms, ks_ms, n=count):
        l2 = workload == "nand_lvl2"
        rotations = n * (2 if workload == "mux" else 1)
        bk_bytes = BK2_BYTES_PER_ROTATION if l2 else BK_BYTES_PER_ROTATION
        kernel = "blind_rotate_lvl2q_kernel" if l2 else "blind_rotate_kernel"
        achieved = bk_bytes * rotations / (br_ms * 1e-3) / 1e9 if br_ms > 0 else 0.0
        facts, note = kernel_facts(kernel)
        r = {
            "bound": "valu_fp64", "kernel": kernel,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "traffic_unit": "bytes per launch",
            "launch_ms": br_ms, "rotations_per_launch": rotations, "algorithmic_bytes_per_rotation": bk_bytes,
            "keyswitch_launch_ms": ks_ms, "pmc_source": note,
        }
        if not l2:
            # SURVEY.md 8(d), secondary: the reference's radix-2 path does 55 296 modular multiplications per CMux step = 34 836 480 per
            # rotation; the rate at which this launch gets through that ALGORITHMIC count (the radix-4 kernels here issue fewer)
            r["reference_modmuls_per_rotation"] = 55296 * 630
            r["algorithmic_modmuls_per_s"] = 55296 * 630 * rotations / (br_ms * 1e-3) if br_ms > 0 else 0.0
        if facts:
            r["traffic"] = facts["hbm_bytes_per_rotation"] * rotations
            r["traffic_replayed_from"] = replayed_from(("fetch", "tcc"))
        r["valu"] = valu_block(kernel, rotations, br_ms, clock_during("lvl2" if l2 else "batch", lambda: run_step(workload, n), br_ms))
        # the number that says how close the kernel is to what bounds it, next to the accounting fraction
        r["frac_valu_fp64"] = r["valu"].get("frac_of_fp64_issue")
        return {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_valu_fp64", "traffic")} | r

    wl = args.workload
    elapsed = distutil.max_over_ranks(timed(wl, args.steps, args.warmup), dist)
    own_s = own["s"]
    br_ms, ks_ms = kernel_times(wl)
    clock_hz = clock_during("lvl2" if wl == "nand_lvl2" else "batch", lambda: run_step(wl), br_ms)      # shader clock under the workload's own launch
    # every rank's own rate, device and launch times, gathered on all ranks (gloo): rank 0 prints them
    reports = distutil.gather_objects({
        "rank": RANK, "local_rank": LOCAL_RANK, "hip_device": DEV, "gpu": identity, "gates_per_step": count,
        "value": count * args.steps / own_s if own_s > 0 else None, "ms_per_step": 1e3 * own_s / args.steps,
        "blind_rotate_launch_ms": br_ms, "keyswitch_launch_ms": ks_ms, "clock_hz": clock_hz}, dist)

    def mixed_strong_extra(total=32768, steps=2, warmup=1):
        """BASELINE configs[2] inside a plain multi-rank run: `total` mixed And/Or/Xor/Nand gates per step, gate i on rank
        floor(i / ceil(total / W)) (test/test_gate_gpu_multi.cc:36-93 splits its gates over the GPUs the same way), same processes,
        same key replicas, no collective in the data path.  Every rank takes part (barriers)."""
        lo, hi = distutil.shard(total, RANK, WORLD)
        cnt = hi - lo
        r = np.random.default_rng(4300 + RANK)
        xin = [r.integers(0, 2**32, size=(max(cnt, 1), ol.n + 1), dtype=np.uint64).astype(np.uint32) for _ in range(3)]
        xd = [api.DeviceBuffer(x.size).upload(x) for x in xin]
        xout = api.DeviceBuffer(max(cnt, 1) * (ol.n + 1))
        xops = np.array([[api.AND, api.OR, api.XOR, api.NAND][(lo + g) % 4] for g in range(max(cnt, 1))], np.int32)

        def step():
            if cnt:
                api.gate_batch(xops, 0, xout, xd[0], xd[1], xd[2], count=cnt, device=0, stream=st.st())
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        eng.Synchronize()
        torch.cuda.synchronize()
        own_t = time.perf_counter() - t0
        barrier()
        el = distutil.max_over_ranks(time.perf_counter() - t0, dist)
        per = distutil.gather_objects({"rank": RANK, "first_gate": lo, "gates_per_step": cnt, "ms_per_step": 1e3 * own_t / steps,
                                       "value": cnt * steps / own_t if cnt and own_t > 0 else None}, dist)
        match = None
        if RANK == 0 and cnt and not args.no_cpu_baseline:
            import oracle_lib                 # the checker, on a sample of rank 0's gates
            L = oracle_lib.load()
            oek = L.orc_evalkey_create(bk, ksk)
            got = xout.download().reshape(max(cnt, 1), ol.n + 1)
            idx = np.arange(1, cnt, max(1, cnt // 16))[:16]
            match = oracle_check(ol, L, oek, xops, xin[0], xin[1], xin[2], got, idx)
            L.orc_evalkey_destroy(oek)
        for b in xd + [xout]:
            b.free()
        return {"value": total * steps / el, "unit": "gates/s", "scaling": "strong", "ms_per_step": 1e3 * el / steps, "steps": steps,
                "warmup": warmup, "total_gates_per_step": total, "baseline_config": "configs[2]", "per_rank": per,
                "gpu_words_match_oracle": match,
                "sharding": "gate i -> rank floor(i / ceil(total / W)), op = (And, Or, Xor, Nand)[i mod 4], per-GPU BK/KSK replica, no collective"}

    mixed_strong = None
    if WORLD > 1 and not strong and wl == "nand" and not args.no_extra:
        mixed_strong = mixed_strong_extra()

    extras_on = RANK == 0 and WORLD == 1 and not args.no_extra and not strong
    latency_ms = None
    if extras_on and not lvl2:
        # ms/gate latency: one gate alone on the idle device, enqueue -> result on the stream
        # (a 1-gate launch takes the workgroup-per-rotation kernel)
        lat = []
        for _ in range(7):
            eng.Synchronize()
            t1 = time.perf_counter()
            run_step("nand", 1)
            eng.Synchronize()
            lat.append(1e3 * (time.perf_counter() - t1))
        latency_ms = sorted(lat)[len(lat) // 2]

    if RANK == 0:
        gates_per_step = args.total_gates if strong else count * WORLD
        res = {
            "metric": "nand_gate_bootstraps_per_sec" if wl == "nand" else f"{wl}_gates_per_sec",
            "value": gates_per_step * args.steps / elapsed,
            "unit": "gate-bootstraps/s",
            "n_gpus": distinct_gpus,       # == WORLD unless ranks share a GPU (--allow-shared-gpu: `shared_gpu`, `ranks`)
            "ranks": WORLD,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"{args.total_gates} {wl.upper()} gates per step split contiguously over {WORLD} GPU(s)" if strong else
                             f"{count} independent {wl.upper()} gates per GPU per step") +
                            f" (BASELINE configs[{dict(nand=1, mux=3, mixed=2, nand_lvl2=4)[wl]}]), " +
                            ("TFHE n=630 N=2048 k=1 l=4 Bgbit=9 64-bit torus, t=7 basebit=2" if lvl2 else
                             "TFHE n=630 N=1024 k=1 l=3 Bgbit=6 t=8 basebit=2") + ", lvl0 ciphertexts resident in HBM",
                "gates_per_gpu": count,
                "sharding": "gates split across ranks, per-GPU BK/KSK replica, no collective",
            },
            "ms_per_gate_throughput": 1e3 * elapsed / (gates_per_step * args.steps) * WORLD,
            **distutil.rank_summary(reports, args.allow_shared_gpu),
            "ms_per_gate_latency_single_gate": latency_ms,
            "roofline": roofline(wl, br_ms, ks_ms),
            "notes": "profiles/bench_line_notes.md",
        }
        if mixed_strong is not None:
            res["extra_workloads"] = {"mixed_32768_strong": mixed_strong}
        main_out = dout.download().reshape(max(count, 1), ol.n + 1)[:count]

        if extras_on and not args.no_api:
            # the reference-style per-gate API, PCIe-inclusive (host-resident ciphertexts, 256 streams,
            # enqueue -> Synchronize): its own process, this one is idle meanwhile
            exe = os.path.join(ROOT, "tools", "bench_api")
            try:
                out = subprocess.run([exe, "4096"], capture_output=True, text=True, timeout=300,
                                     env=dict(os.environ, HIP_VISIBLE_DEVICES=str(DEV)) if WORLD > 1 else None)
                lines = [json.loads(l) for l in out.stdout.strip().splitlines() if l.startswith("{")]
                res["api_pcie_inclusive"] = lines[0]
                # enqueue -> Synchronize with H2D/D2H of the ciphertexts inside the timed region
                # (the reference's own way of timing, test/test_util.h:29-72); `value` above is the inputs-resident rate
                res["value_pcie_inclusive"] = lines[0].get("gates_per_s")
                for ln in lines[1:]:         # the netlist lines say what they are
                    if "single-assignment" in ln.get("netlist", ""):
                        res["api_pcie_inclusive"]["netlist_level_reassignment_bound"] = ln
                    elif ln.get("sched_rename") == 0:
                        res["api_pcie_inclusive"]["depth_first_netlist_without_renaming"] = ln
                    elif ln.get("sched_two_lane") == 0:
                        res["api_pcie_inclusive"]["depth_first_netlist_level_by_level"] = ln
                    else:
                        res["api_pcie_inclusive"]["depth_first_netlist"] = ln
                # one cufhe::Nand on host ciphertexts, call -> Synchronize, and a chain of 16 dependent ones (tools/api_latency.cpp): the
                # ms/gate a user of the reference's API sees (test/test_api_gpu.cu:140-159 is the chained pattern)
                lat = subprocess.run([os.path.join(ROOT, "tools", "api_latency")], capture_output=True, text=True, timeout=120,
                                     env=dict(os.environ, HIP_VISIBLE_DEVICES=str(DEV)) if WORLD > 1 else None)
                ones = [float(l.split("call + Synchronize")[1].split("ms")[0]) for l in lat.stdout.splitlines() if l.startswith("one Nand")]
                chains = [float(l.split("=")[1].split("ms")[0]) for l in lat.stdout.splitlines() if l.startswith("chain of 16")]
                if ones and chains:
                    res["api_pcie_inclusive"]["single_nand_call_to_synchronize_ms"] = min(ones)
                    res["api_pcie_inclusive"]["chain_of_16_dependent_nand_ms_per_gate"] = min(chains)
            except Exception as e:      # the figure is auxiliary: never lose the headline line to it
                res["api_pcie_inclusive"] = {"error": repr(e)}

        if not args.no_cpu_baseline and WORLD == 1 and not strong:
            import oracle_lib                 # the only leg that touches oracle/: checker and timed CPU baseline
            L = oracle_lib.load()
            oek = L.orc_evalkey_create(bk, ksk)
            if wl == "nand":
                res["cpu_baseline"] = cpu_baseline(ol, L, bk, ksk, in0, in1, main_out, oek)
            elif wl in ("mux", "mixed"):
                idx = np.arange(1, count, max(1, count // 16))[:16]
                ops = {"mux": api.MUX, "mixed": mixed_ops}[wl]
                res["gpu_words_match_oracle"] = oracle_check(ol, L, oek, ops, in0, in1, in2, main_out, idx)
            if extras_on and wl == "nand":
                extra = {}
                for w2 in ("mux", "mixed"):
                    dt = timed(w2, 2, 1)
                    b2, k2 = kernel_times(w2, 1)
                    out2 = dout.download().reshape(count, ol.n + 1)
                    idx = np.arange(1, count, max(1, count // 16))[:16]
                    ops = {"mux": api.MUX, "mixed": mixed_ops}[w2]
                    rf = roofline(w2, b2, k2)
                    extra[w2] = {"value": count * 2 / dt, "unit": "gates/s", "ms_per_step": 1e3 * dt / 2,
                                 "blind_rotate_launch_ms": b2, "keyswitch_launch_ms": k2, "roofline_frac": rf["frac"],
                                 "bound": rf["bound"], "valu": rf["valu"],
                                 "baseline_config": {"mux": "configs[3]", "mixed": "configs[2] op mix on one GPU"}[w2],
                                 "gpu_words_match_oracle": oracle_check(ol, L, oek, ops, in0, in1, in2, out2, idx)}
                try:
                    init_lvl2()
                    dt = timed("nand_lvl2", 2, 1)
                    b2, k2 = kernel_times("nand_lvl2", 1)
                    out2 = dout.download().reshape(count, ol.n + 1)
                    ek2 = L.orc2_evalkey_create(bk2, ksk2)
                    idx = np.arange(1, count, max(1, count // 8))[:8]
                    want = np.zeros(idx.size * (ol.n + 1), np.uint32)
                    a_s, b_s = np.ascontiguousarray(in0[idx]).ravel(), np.ascontiguousarray(in1[idx]).ravel()
                    L.orc2_gate_batch(ek2, np.array([0], np.int32), 0, idx.size, want, a_s, b_s.ctypes.data, None,
                                      min(L.orc_max_threads(), 32))
                    L.orc2_evalkey_destroy(ek2)
                    rf = roofline("nand_lvl2", b2, k2)
                    extra["nand_lvl2"] = {"value": count * 2 / dt, "unit": "gate-bootstraps/s", "ms_per_step": 1e3 * dt / 2,
                                          "blind_rotate_launch_ms": b2, "keyswitch_launch_ms": k2, "roofline_frac": rf["frac"],
                                          "bound": rf["bound"], "valu": rf["valu"],
                                          "roofline_traffic": rf["traffic"], "baseline_config": "configs[4]",
                                          "gpu_words_match_oracle": bool(np.array_equal(want.reshape(idx.size, -1), out2[idx]))}
                except Exception as e:
                    extra["nand_lvl2"] = {"error": repr(e)}
                # the reference's primary tested ciphertext level (test/test_gate_gpu.cc:36-91): NAND on lvl1 ciphertexts, key switch
                # first, then the blind rotation (src/bootstrap_gpu.cu:383-400)
                try:
                    w1 = ol.N + 1
                    r1 = np.random.default_rng(4711)
                    x1 = [r1.integers(0, 2**32, size=(count, w1), dtype=np.uint64).astype(np.uint32) for _ in range(2)]
                    xd1 = [api.DeviceBuffer(x.size).upload(x) for x in x1]
                    xo1 = api.DeviceBuffer(count * w1)

                    def step1():
                        api.gate_batch(api.NAND, 1, xo1, xd1[0], xd1[1], None, count=count, device=0, stream=st.st())
                    step1()
                    eng.Synchronize()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        step1()
                    eng.Synchronize()
                    dt = (time.perf_counter() - t0) / 2
                    got1 = xo1.download().reshape(count, w1)
                    idx = np.arange(1, count, max(1, count // 8))[:8]
                    want1 = np.zeros(idx.size * w1, np.uint32)
                    a_s, b_s = np.ascontiguousarray(x1[0][idx]).ravel(), np.ascontiguousarray(x1[1][idx]).ravel()
                    L.orc_gate_batch(oek, np.array([0], np.int32), 0, 1, idx.size, want1, a_s, b_s.ctypes.data, None, min(L.orc_max_threads(), 32))
                    extra["nand_level1"] = {"value": count / dt, "unit": "gate-bootstraps/s", "ms_per_step": 1e3 * dt,
                                            "what": f"{count} NAND on lvl1 ciphertexts (N + 1 = {w1} words): IdentityKeySwitch, blind rotation, sample extract",
                                            "gpu_words_match_oracle": bool(np.array_equal(want1.reshape(idx.size, w1), got1[idx]))}
                    for b in xd1 + [xo1]:
                        b.free()
                except Exception as e:
                    extra["nand_level1"] = {"error": repr(e)}
                # a mid-sized launch (512 gates: one round of the paired low-latency kernel) and the key switch of the main batch
                try:
                    run_step("nand", 512)
                    eng.Synchronize()
                    api.profile_get(reset=True)
                    api.profile_enable(True)
                    for _ in range(3):
                        run_step("nand", 512)
                    eng.Synchronize()
                    mp = api.profile_get(reset=True)
                    api.profile_enable(False)
                    mbr = mp.blind_rotate_ms / max(mp.blind_rotate_launches, 1)
                    extra["nand_512"] = {"ms_per_step": mbr + mp.keyswitch_ms / max(mp.keyswitch_launches, 1),
                                         "blind_rotate_launch_ms": mbr, "kernel": "blind_rotate_ll2_kernel", "bound": "valu_fp64",
                                         "valu": valu_block("blind_rotate_ll2_kernel", 512, mbr,
                                                            clock_during("ll2", lambda: [run_step("nand", 512) for _ in range(4)], 4 * mbr))}
                except Exception as e:
                    extra["nand_512"] = {"error": repr(e)}
                try:
                    if count > 1600:        # launch_keyswitch: above 1600 ciphertexts the shared-table kernel (capi.hip, kKsAutoWg)
                        kfacts, knote = kernel_facts("keyswitch_kernel")
                        per_wg = -(-count // 256)
                        extra["keyswitch_%d" % count] = {
                            "launch_ms": ks_ms, "kernel": "keyswitch_kernel", "bound": "lds", "ciphertexts": count,
                            "ciphertexts_per_workgroup": min(16, per_wg),
                            "algorithmic_bytes": count * 15507456,
                            # the PMC facts were recorded on launches of 4096 ciphertexts (16 per workgroup): only then do they describe this launch
                            "lds_pipe_busy_under_profiler": kfacts.get("lds_pipe_busy") if kfacts and count == 4096 else None,
                            "valu_pipe_busy_under_profiler": kfacts.get("valu_pipe_busy") if kfacts and count == 4096 else None,
                            "lds_bank_conflict_frac": kfacts.get("lds_bank_conflict_frac") if kfacts and count == 4096 else None,
                            "pmc_source": knote, "replayed_from": replayed_from(("sq", "lds"))}
                except Exception as e:
                    extra["keyswitch_%d" % count] = {"error": repr(e)}
                # the other compiled parameter sets through the generic kernels (kernels_ps.hip.h),
                # 4096 NAND on random keys, sampled words against the oracle compiled for the set
                psets = {}
                for ps in range(api.ps_count()):
                    pp = api.ps_params(ps)
                    name = pp.name.decode()
                    try:
                        prng = np.random.default_rng(100 + ps)
                        pbk = prng.integers(0, 2**32, size=int(pp.bk_words), dtype=np.uint64).astype(np.uint32)
                        pksk = prng.integers(0, 2**32, size=int(pp.ksk_words), dtype=np.uint64).astype(np.uint32)
                        api.ps_initialize(ps, pbk, pksk)
                        w = int(pp.lvl0_words)
                        pin = [prng.integers(0, 2**32, size=(count, w), dtype=np.uint64).astype(np.uint32) for _ in range(2)]
                        pd = [api.DeviceBuffer(count * w).upload(x) for x in pin]
                        pout = api.DeviceBuffer(count * w)

                        def ps_step():
                            api.ps_gate_batch(ps, api.NAND, pout, pd[0], pd[1], count=count, stream=st.st())
                        ps_step()
                        eng.Synchronize()
                        t0 = time.perf_counter()
                        for _ in range(2):
                            ps_step()
                        eng.Synchronize()
                        dt = (time.perf_counter() - t0) / 2
                        # the blind-rotate launch alone, HIP events on the launch stream
                        api.profile_get(reset=True)
                        api.profile_enable(True)
                        ps_step()
                        eng.Synchronize()
                        pprof = api.profile_get(reset=True)
                        api.profile_enable(False)
                        pbr_ms = pprof.blind_rotate_ms / max(pprof.blind_rotate_launches, 1)
                        got = pout.download().reshape(count, w)
                        Ls = oracle_lib.load_set(name)
                        eks = Ls.orc_evalkey_create(pbk, pksk)
                        idx = np.arange(1, count, max(1, count // 8))[:8]
                        want = np.zeros(idx.size * w, np.uint32)
                        a_s, b_s = np.ascontiguousarray(pin[0][idx]).ravel(), np.ascontiguousarray(pin[1][idx]).ravel()
                        Ls.orc_gate_batch(eks, np.array([0], np.int32), 0, 0, idx.size, want, a_s, b_s.ctypes.data, None,
                                          min(Ls.orc_max_threads(), 32))
                        Ls.orc_evalkey_destroy(eks)
                        bk_bytes = int(pp.bk_ntt_bytes)        # n (k+1)^2 l N 8 x key limbs: one sweep of the NTT-domain key
                        psets[name] = {"params": f"n={pp.n} N={pp.N} k={pp.k} l={pp.l} Bgbit={pp.Bgbit} key_limbs={pp.key_limbs}",
                                       "value": count / dt, "unit": "gate-bootstraps/s", "ms_per_step": 1e3 * dt,
                                       "kernel": "blind_rotate_ps_batch_kernel<PS> (wave per rotation, written once over the set; not hand-scheduled)",
                                       "blind_rotate_launch_ms": pbr_ms,
                                       "bk_sweep_frac_of_hbm_peak": count * bk_bytes / dt / 8e12,
                                       "bound": "valu_fp64",
                                       "valu": valu_block("blind_rotate_ps_batch_kernel<%s>" % name, count, pbr_ms,
                                                          clock_during("ps_" + name, ps_step, pbr_ms)),
                                       "gpu_words_match_oracle": bool(np.array_equal(want.reshape(idx.size, w), got[idx]))}
                    except Exception as e:
                        psets[name] = {"error": repr(e)}
                extra["param_sets"] = psets
                res["extra_workloads"] = extra
            L.orc_evalkey_destroy(oek)
        if lvl2 and not args.no_cpu_baseline and WORLD == 1:
            import oracle_lib
            L = oracle_lib.load()
            ek2 = L.orc2_evalkey_create(bk2, ksk2)
            threads = min(L.orc_max_threads(), 32)
            n_chk = 2 * threads
            want = np.zeros(n_chk * (ol.n + 1), np.uint32)
            t0 = time.perf_counter()
            a_s, b_s = np.ascontiguousarray(in0[:n_chk]).ravel(), np.ascontiguousarray(in1[:n_chk]).ravel()
            L.orc2_gate_batch(ek2, np.array([0], np.int32), 0, n_chk, want, a_s, b_s.ctypes.data, None, threads)
            dt = time.perf_counter() - t0
            L.orc2_evalkey_destroy(ek2)
            res["cpu_baseline"] = {"value": n_chk / dt, "unit": "gate-bootstraps/s", "cores": int(threads), "kind": "port",
                                   "implementation": "oracle/tfhe_oracle_lvl2.c (the checker itself; no optimised CPU path for this ring)",
                                   "sample": f"{n_chk} of the batch's NAND gates, {dt:.1f} s",
                                   "gpu_words_match_oracle": bool(np.array_equal(want.reshape(n_chk, -1), main_out[:n_chk]))}
        print(json.dumps(ordered_line(res)), file=json_out, flush=True)

    st.Destroy()
    eng.CleanUp()
    if WORLD > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
