#!/usr/bin/env python3
"""kernel_facts.py -- turn rocprofv3 PMC passes of `bench.py` into profiles/kernel_facts.json.

bench.py prices its launches with the instruction counts and HBM bytes per rotation recorded here (properties of the
code: they do not depend on the box), and only while every file with device code still hashes to `source_sha256`;
clocks and launch times always come from the bench run itself.

Usage (after the PMC passes of tools/profile_round.sh -- separate runs, --pmc with --kernel-trace only):
    python tools/kernel_facts.py --tag r03 --label "..." gpurun_out/r03_prof
Every argument is a directory (searched recursively) or a *_counter_collection.csv file.  Corrections applied as
MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE is in KB and counts half of a wide coalesced streaming read on gfx950
(x2); WRITE_SIZE is exact; GRBM_GUI_ACTIVE is the sum over the 8 XCDs.
"""
import argparse
import csv
import glob
import importlib.util
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STEPS = 630
# key in kernel_facts.json -> (substring that must be in the kernel name, substrings that must not, rotations (units) per
# workgroup, waves per workgroup, waves per SIMD, CMux steps per unit or None)
KERNELS = {
    "blind_rotate_kernel": ("blind_rotate_kernel", (), 8, 8, 2, STEPS),
    "blind_rotate_lvl2_kernel": ("blind_rotate_lvl2_kernel", (), 1, 8, 2, STEPS),
    "blind_rotate_lvl2q_kernel": ("blind_rotate_lvl2q_kernel", (), 1, 4, 2, STEPS),
    "blind_rotate_ll2_kernel": ("blind_rotate_ll2_kernel", (), 2, 16, 4, STEPS),
    "blind_rotate_ll_kernel": ("blind_rotate_ll_kernel", (), 1, 16, 4, STEPS),
    "blind_rotate_ps_batch_kernel<default>": ("blind_rotate_ps_batch_kernel", ("K2N512", "Cggi16", "SmallMod"), 8, 8, 2, STEPS),
    "blind_rotate_ps_batch_kernel<smallmod>": ("blind_rotate_ps_batch_kernel<cufhe_amd::PsSmallMod", (), 8, 8, 2, STEPS),
    "blind_rotate_ps_batch_kernel<k2n512>": ("blind_rotate_ps_batch_kernel<cufhe_amd::PsK2N512", (), 8, 8, 2, STEPS),
    "blind_rotate_ps_batch_kernel<cggi16>": ("blind_rotate_ps_batch_kernel<cufhe_amd::PsCggi16", (), 8, 8, 2, 500),
    # units per workgroup: 16 for the launches of 4096 ciphertexts (capi.hip: ks_auto_shape); the smaller launches of the profiled
    # command fill the same grid with runs of j and are left out by their duration (main() below)
    "keyswitch_kernel": ("keyswitch_kernel<cufhe_amd::KsShapeDefault", (), 16, 16, 4, None),
    # the lvl20 shape always cuts j into at least two runs: 4096 ciphertexts are 512 workgroups of 16
    "keyswitch_kernel<lvl2>": ("keyswitch_kernel<cufhe_amd::KsShapeLvl2", (), 8, 16, 4, None),      # 16 ciphertexts per TWO workgroups
}


def bench_module_hash():
    """the hash bench.py computes over the device sources (loaded from bench.py itself: one definition)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    ns = {"os": os, "hashlib": __import__("hashlib"), "ROOT": ROOT}
    start = src.index("DEVICE_SOURCES = ")
    end = src.index("_FACTS = None")
    exec(src[start:end], ns)
    return ns["source_hash"]()


def which(name):
    for key, (must, must_not, *_rest) in KERNELS.items():
        if must in name and not any(x in name for x in must_not):
            if key == "blind_rotate_kernel" and any(x in name for x in ("lvl2", "_ll_", "_ll2_", "_wg_", "_ps_")):
                continue
            return key
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--label", default="")
    ap.add_argument("paths", nargs="+")
    args = ap.parse_args()
    files = []
    for p in args.paths:
        if os.path.isdir(p):
            files += glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
        else:
            files.append(p)
    # counter -> kernel -> list of (grid, wg, value, duration_ns)
    vals = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = which(r["Kernel_Name"])
            if not k:
                continue
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            vals.setdefault(r["Counter_Name"], {}).setdefault(k, []).append(
                (int(r["Grid_Size"]), int(r["Workgroup_Size"]), float(r["Counter_Value"]), dur))
    out = {"source_sha256": bench_module_hash(), "recorded": args.label or args.tag,
           "sources": sorted({f"{args.tag}_{os.path.basename(os.path.dirname(f))}_counter_collection.csv" for f in files}), "kernels": {}}
    # one launch shape per row: the largest-grid launches of a kernel must have one duration (a launch with fewer units per
    # workgroup hides behind the same grid -- the key switch with 2048 ciphertexts -- and would be averaged in)
    for k in KERNELS:
        rows = vals.get("SQ_INSTS_VALU", {}).get(k)
        if rows:
            gmax = max(r[0] for r in rows)
            durs = sorted(r[3] for r in rows if r[0] == gmax)
            if k.startswith("keyswitch_kernel"):
                durs = [x for x in durs if x >= 0.85 * durs[-1]]
            if durs[-1] > 1.25 * durs[0]:
                print(f"WARNING {k}: launches of the largest grid differ in duration ({durs[0] * 1e-6:.2f} .. {durs[-1] * 1e-6:.2f} ms): "
                      "more than one launch shape in this row?", file=sys.stderr)
    for k, (_m, _n, rot_per_wg, waves_per_wg, waves_per_simd, steps) in KERNELS.items():
        def avg(counter):
            rows = vals.get(counter, {}).get(k)
            if not rows:
                return None, None, None
            gmax = max(r[0] for r in rows)           # the largest launch shape of this kernel in the run
            rows = [r for r in rows if r[0] == gmax]
            # keyswitch_kernel fills one grid round at every launch size (fewer ciphertexts per workgroup, or fewer steps of j per
            # workgroup): the launches of 4096 ciphertexts -- 16 per workgroup, all 1024 steps -- are the slowest of that grid
            if k.startswith("keyswitch_kernel"):
                dmax = max(r[3] for r in rows)
                rows = [r for r in rows if r[3] >= 0.85 * dmax]
            return sum(r[2] for r in rows) / len(rows), gmax // rows[0][1], sum(r[3] for r in rows) / len(rows)
        insts, wgs, dur_ns = avg("SQ_INSTS_VALU")
        if insts is None:
            continue
        rotations = wgs * rot_per_wg
        gui, _, gui_dur = avg("GRBM_GUI_ACTIVE")
        clock = gui / 8.0 / (gui_dur * 1e-9) if gui else None
        fetch, _, _ = avg("FETCH_SIZE")
        write, _, _ = avg("WRITE_SIZE")
        hit, _, _ = avg("TCC_HIT_sum")
        miss, _, _ = avg("TCC_MISS_sum")
        hbm = (2.0 * fetch + (write or 0.0)) * 1024.0 if fetch is not None else None
        k_out = {
            "rotations_per_launch": rotations,
            "launch_ms_under_profiler": dur_ns * 1e-6,
            "valu_insts_per_launch": insts,
            "valu_insts_per_rotation": insts / rotations,
            "valu_insts_per_step_per_wave": insts / (wgs * waves_per_wg * steps) if steps else None,
            "clock_hz_under_profiler": clock,
            "valu_pipe_busy": 4.0 * insts / (1024 * clock * dur_ns * 1e-9) if clock else None,
            "waves_per_simd": waves_per_simd,
            "hbm_bytes_per_launch": hbm,
            "hbm_bytes_per_rotation": hbm / rotations if hbm is not None else None,
            "fetch_size_kb_raw": fetch, "write_size_kb": write,
            "tcc_hit_rate": hit / (hit + miss) if hit is not None and miss is not None and hit + miss > 0 else None,
        }
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "SQ_INSTS_SALU",
                  "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_LDS_ADDR_CONFLICT",
                  "SQ_INSTS_VMEM_RD", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES"):
            v, _, _ = avg(c)
            if v is not None:
                k_out[c] = v
        if k_out.get("SQ_LDS_IDX_ACTIVE"):
            k_out["lds_bank_conflict_frac"] = k_out.get("SQ_LDS_BANK_CONFLICT", 0.0) / k_out["SQ_LDS_IDX_ACTIVE"]
            if clock:      # LDS-array cycles over the cycles of 256 CUs in the launch (the counter sums over CUs)
                k_out["lds_pipe_busy"] = k_out["SQ_LDS_IDX_ACTIVE"] / (256 * clock * dur_ns * 1e-9)
        out["kernels"][k] = k_out
    dst = os.path.join(ROOT, "profiles", "kernel_facts.json")
    json.dump(out, open(dst, "w"), indent=1)
    for f in files:      # keep the raw passes beside it, named per round and per pass
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{args.tag}_{os.path.basename(os.path.dirname(f))}_counter_collection.csv"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
