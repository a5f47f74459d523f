#!/usr/bin/env python3
"""bench.py -- NAND gate-bootstraps/s of the HIP gate path (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: 4096 independent NAND gates
(BASELINE.json configs[1]) per GPU, inputs and keys already resident in HBM, one
cufhe_amd_gate_batch call = one blind-rotate launch + one key-switch launch.  With N > 1
ranks (one per GPU, launched by torch.distributed.run) every rank runs its own 4096 gates
against its own BK/KSK replica: weak scaling, no data-path collective.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (blind rotate)
with the BK-sweep accounting of SURVEY.md 8(d): 61 931 520 algorithmic bytes per rotation,
divided by the launch time measured with HIP events on the launch stream.  `cpu_baseline`
times the CPU oracle (oracle/, a restatement of the same gate) on this box's host cores on
a bounded sample of the same inputs and checks the GPU words against it.
"""
import argparse
import json
import os
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

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: its bundled HIP runtime is the one the process uses)
import torch.distributed as dist  # noqa: E402

BK_BYTES_PER_ROTATION = 61931520          # n (k+1)^2 l N 8, SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8 TB/s spec


def recorded_hbm_traffic(rotations, lvl2=False):
    """HBM bytes per blind-rotate launch from the committed rocprofv3 PMC passes (separate
    --pmc runs of this same command, profiles/r01_{final,lvl2}_pmc_*): FETCH_SIZE is in KB and reads
    half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact.
    Only meaningful for the launch shape it was recorded on (4096 rotations); else None."""
    import csv
    if rotations != 4096:
        return None, None
    tag, kernel = ("r01_lvl2", "blind_rotate_lvl2_kernel") if lvl2 else ("r01_final", "blind_rotate_kernel")
    try:
        vals = {}
        for name, f in (("FETCH_SIZE", f"{tag}_pmc_fetch_counter_collection.csv"),
                        ("WRITE_SIZE", f"{tag}_pmc_tcc_counter_collection.csv")):
            path = os.path.join(ROOT, "profiles", f)
            best = 0.0
            for r in csv.DictReader(open(path)):
                if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
                    best = max(best, float(r["Counter_Value"]))
            vals[name] = best
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, f"profiles/{tag}_pmc_{{fetch,tcc}}_counter_collection.csv"
    except Exception:
        return None, None


def cpu_baseline(eng, ol, bk, ksk, in0, in1, gpu_out, target_seconds=15.0):
    """Time the CPU oracle on a bounded sample of the same workload; verify GPU words."""
    import ctypes
    L = ol.load()
    ek = L.orc_evalkey_create(bk, ksk)
    threads = L.orc_max_threads()
    nand = np.array([0], np.int32)
    words = ol.n + 1

    def run(count):
        out = np.zeros(count * words, np.uint32)
        a = np.ascontiguousarray(in0[:count]).ravel()
        b = np.ascontiguousarray(in1[:count])
        t = time.perf_counter()
        L.orc_gate_batch(ek, nand, 0, 0, count, out, a, b.ctypes.data, None, threads)
        return time.perf_counter() - t, out.reshape(count, words)

    # the container's CPU share can be far below the visible core count: calibrate the
    # thread count on one round each and keep the fastest
    best = None
    for cand in sorted({min(16, threads), min(32, threads), threads}):
        threads = cand
        dt, _ = run(cand)
        if best is None or cand / dt > best[0]:
            best = (cand / dt, cand, dt)
    _, threads, dt = best
    per_round = max(dt, 1e-3)
    count = int(min(in0.shape[0], max(threads, threads * round(target_seconds / per_round))))
    dt, out = run(count)
    L.orc_evalkey_destroy(ek)
    match = bool(np.array_equal(out, gpu_out[:count]))
    return {
        "value": count / dt, "unit": "gate-bootstraps/s", "cores": int(threads), "kind": "port",
        "visible_cores": int(L.orc_max_threads()),
        "sample": f"{count} of the batch's NAND gates, OpenMP over gates, {dt:.1f} s",
        "ms_per_gate_per_core": 1e3 * dt * threads / count,
        "gpu_words_match_oracle": match,
    }


def cpu_baseline_lvl2(ol, bk, ksk, in0, in1, gpu_out, target_seconds=15.0):
    """Same for the N = 2048 workload (oracle/tfhe_oracle_lvl2.c)."""
    L = ol.load()
    ek = L.orc2_evalkey_create(bk, ksk)
    threads = L.orc_max_threads()
    nand = np.array([0], np.int32)
    words = ol.n + 1

    def run(count, threads):
        out = np.zeros(count * words, np.uint32)
        a = np.ascontiguousarray(in0[:count]).ravel()
        b = np.ascontiguousarray(in1[:count])
        t = time.perf_counter()
        L.orc2_gate_batch(ek, nand, 0, count, out, a, b.ctypes.data, None, threads)
        return time.perf_counter() - t, out.reshape(count, words)

    threads = min(threads, 32)
    dt, _ = run(threads, threads)
    count = int(min(in0.shape[0], max(threads, threads * round(target_seconds / max(dt, 1e-3)))))
    dt, out = run(count, threads)
    L.orc2_evalkey_destroy(ek)
    return {
        "value": count / dt, "unit": "gate-bootstraps/s", "cores": int(threads), "kind": "port",
        "visible_cores": int(L.orc_max_threads()),
        "sample": f"{count} of the batch's NAND gates, OpenMP over gates, {dt:.1f} s",
        "ms_per_gate_per_core": 1e3 * dt * threads / count,
        "gpu_words_match_oracle": bool(np.array_equal(out, gpu_out[:count])),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gates", type=int, default=4096, help="gates per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency", action="store_true",
                    help="also time one gate alone on the idle device (adds 1-gate launches to a profile)")
    ap.add_argument("--workload", choices=["nand", "mux", "mixed", "nand_lvl2"], default="nand",
                    help="nand = BASELINE configs[1] (the metric's config); mux = configs[3]; mixed = configs[2] op mix; "
                         "nand_lvl2 = configs[4] (N = 2048 ring, 64-bit torus)")
    args = ap.parse_args()
    if args.gpus != WORLD:
        if WORLD == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")

    if WORLD > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=RANK, world_size=WORLD)
    DEV = distutil.device_for_rank(LOCAL_RANK, torch.cuda.device_count()) if WORLD > 1 else 0
    torch.cuda.set_device(DEV)
    torch.cuda.init()

    import cufhe_amd as eng                  # fails loudly if the HIP library is missing
    eng.api.set_option("device_base", DEV)   # logical device 0 of this process = this rank's GPU

    class ol:                                 # sizes come from the library, not from the oracle
        n, N = int(eng.PARAMS.n), int(eng.PARAMS.N)
        BK_WORDS, KSK_WORDS = int(eng.PARAMS.bk_words), int(eng.PARAMS.ksk_words)

    count = args.gates
    rng = np.random.default_rng(42 + RANK)
    # synthetic keys and ciphertexts: uniform torus words (the path's work is data-independent)
    lvl2 = args.workload == "nand_lvl2"
    if lvl2:
        p2 = eng.lvl2_params()
        bk = rng.integers(0, 2**64, size=int(p2.bk_words), dtype=np.uint64)
        ksk = rng.integers(0, 2**32, size=int(p2.ksk_words), dtype=np.uint64).astype(np.uint32)
    else:
        bk = rng.integers(0, 2**32, size=ol.BK_WORDS, dtype=np.uint64).astype(np.uint32)
        ksk = rng.integers(0, 2**32, size=ol.KSK_WORDS, dtype=np.uint64).astype(np.uint32)
    in0 = rng.integers(0, 2**32, size=(count, ol.n + 1), dtype=np.uint64).astype(np.uint32)
    in1 = rng.integers(0, 2**32, size=(count, ol.n + 1), dtype=np.uint64).astype(np.uint32)

    eng.SetGPUNum(1)
    if lvl2:
        eng.lvl2_initialize(bk, ksk)
    else:
        eng.Initialize(bk, ksk)
    d0 = eng.api.DeviceBuffer(in0.size).upload(in0)
    d1 = eng.api.DeviceBuffer(in1.size).upload(in1)
    in2 = rng.integers(0, 2**32, size=(count, ol.n + 1), dtype=np.uint64).astype(np.uint32)
    d2 = eng.api.DeviceBuffer(in2.size).upload(in2)
    dout = eng.api.DeviceBuffer(count * (ol.n + 1))
    if args.workload in ("nand", "nand_lvl2"):
        ops, rot_per_gate = eng.api.NAND, 1
    elif args.workload == "mux":
        ops, rot_per_gate = eng.api.MUX, 2
    else:   # configs[2]: op[i] = {AND, OR, XOR, NAND}[i mod 4]
        ops = np.array([[eng.api.AND, eng.api.OR, eng.api.XOR, eng.api.NAND][g % 4] for g in range(count)], np.int32)
        rot_per_gate = 1
    st = eng.Stream(0)
    st.Create()

    def step():
        if lvl2:
            eng.lvl2_gate_batch(ops, dout, d0, d1, d2, count=count, device=0, stream=st.st())
        else:
            eng.gate_batch(ops, 0, dout, d0, d1, d2, count=count, device=0, stream=st.st())

    def barrier():
        if WORLD > 1:
            dist.barrier()
        eng.Synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_get(reset=True)
    eng.profile_enable(False)
    elapsed = distutil.max_over_ranks(elapsed, dist)

    latency_ms = None
    if RANK == 0 and args.latency and not lvl2:
        # ms/gate latency: one gate alone on the idle device, enqueue -> result on the stream
        # (a 1-gate launch takes the workgroup-per-rotation kernel)
        lat = []
        for _ in range(5):
            eng.Synchronize()
            t1 = time.perf_counter()
            eng.gate_batch(eng.api.NAND, 0, dout, d0, d1, count=1, device=0, stream=st.st())
            eng.Synchronize()
            lat.append(1e3 * (time.perf_counter() - t1))
        latency_ms = sorted(lat)[len(lat) // 2]
    if RANK == 0:
        total_gates = count * args.steps * WORLD
        br_ms = prof.blind_rotate_ms / max(prof.blind_rotate_launches, 1)
        ks_ms = prof.keyswitch_ms / max(prof.keyswitch_launches, 1)
        rotations = count * rot_per_gate
        # lvl2: the same n (k+1)^2 l N 8 accounting at N = 2048, l = 4 (the device key is 3x that: three limbs)
        bk_bytes = 630 * 4 * 4 * 2048 * 8 if lvl2 else BK_BYTES_PER_ROTATION
        achieved = bk_bytes * rotations / (br_ms * 1e-3) / 1e9
        traffic, traffic_src = recorded_hbm_traffic(rotations, lvl2)
        res = {
            "metric": "nand_gate_bootstraps_per_sec" if args.workload == "nand" else f"{args.workload}_gates_per_sec",
            "value": total_gates / elapsed,
            "unit": "gate-bootstraps/s",
            "n_gpus": WORLD,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{count} independent {args.workload.upper()} gates per GPU per step "
                            f"(BASELINE configs[{dict(nand=1, mux=3, mixed=2, nand_lvl2=4)[args.workload]}]), " +
                            ("TFHE n=630 N=2048 k=1 l=4 Bgbit=9 64-bit torus, t=7 basebit=2" if lvl2 else
                             "TFHE n=630 N=1024 k=1 l=3 Bgbit=6 t=8 basebit=2") + ", lvl0 ciphertexts resident in HBM",
                "gates_per_gpu": count,
                "sharding": "gates split across ranks, per-GPU BK/KSK replica, no collective",
            },
            "ms_per_gate_throughput": 1e3 * elapsed / (count * args.steps),
            "ms_per_gate_latency_single_gate": latency_ms,
            "roofline": {
                "bound": "hbm", "kernel": "blind_rotate_lvl2_kernel" if lvl2 else "blind_rotate_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                "traffic_source": traffic_src,
                "launch_ms": br_ms, "rotations_per_launch": rotations,
                "algorithmic_bytes_per_rotation": bk_bytes,
                "keyswitch_launch_ms": ks_ms,
            },
        }
        if not args.no_cpu_baseline and WORLD == 1 and args.workload == "nand":
            gpu_out = dout.download().reshape(count, ol.n + 1)
            import oracle_lib                 # the only leg that touches the CPU oracle
            res["cpu_baseline"] = cpu_baseline(eng, oracle_lib, bk, ksk, in0, in1, gpu_out)
        if not args.no_cpu_baseline and WORLD == 1 and lvl2:
            gpu_out = dout.download().reshape(count, ol.n + 1)
            import oracle_lib
            res["cpu_baseline"] = cpu_baseline_lvl2(oracle_lib, bk, ksk, in0, in1, gpu_out)
        print(json.dumps(res), flush=True)

    st.Destroy()
    eng.CleanUp()
    if WORLD > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
