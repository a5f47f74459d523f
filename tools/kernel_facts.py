#!/usr/bin/env python3
"""kernel_facts.py -- turn rocprofv3 PMC passes of `bench.py` into profiles/kernel_facts.json.

bench.py prints `roofline.traffic` and `roofline.valu` from this file, and only while
cufhe_amd/csrc still hashes to `source_sha256`: counters recorded for other code are never shown.

Usage (after the three PMC passes of the guide's recipe -- separate runs, --pmc with --kernel-trace only):
    python tools/kernel_facts.py --tag r02 --label "..." gpurun_out/pmc_sq gpurun_out/pmc_fetch gpurun_out/pmc_tcc [...]
Every argument is a directory (searched recursively) or a *_counter_collection.csv file.  Corrections
applied as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE is in KB and counts half of a wide coalesced
streaming read on gfx950 (x2); WRITE_SIZE is exact; GRBM_GUI_ACTIVE is the sum over the 8 XCDs.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERNELS = {
    # name -> (rotations per workgroup, waves per workgroup, CMux steps)
    "blind_rotate_kernel": dict(rot_per_wg=8, waves_per_wg=8, waves_per_simd=2),
    "blind_rotate_lvl2_kernel": dict(rot_per_wg=1, waves_per_wg=8, waves_per_simd=2),
}
STEPS = 630


def source_hash():
    """the same hash bench.py computes: sha256 over cufhe_amd/csrc/*.{h,hip}"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cufhe_amd", "csrc")
    # the device code of the two profiled kernels and what they include (host-side files do not change what a launch executes)
    for f in ("fpfield.h", "ntt_wave.h", "kernels_common.hip.h", "kernels.hip.h", "kernels_lvl2.hip.h"):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02")
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
            name = r["Kernel_Name"].split("(")[0]
            for k in KERNELS:
                if name.endswith(k) or name == k or (k in name and "lvl2" not in name.replace(k, "")):
                    if k == "blind_rotate_kernel" and ("lvl2" in name or "_ll_" in name or "_wg_" in name):
                        continue
                    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    vals.setdefault(r["Counter_Name"], {}).setdefault(k, []).append(
                        (int(r["Grid_Size"]), int(r["Workgroup_Size"]), float(r["Counter_Value"]), dur))
    out = {"source_sha256": source_hash(), "recorded": args.label or args.tag, "kernels": {}}
    for k, shape in KERNELS.items():
        def avg(counter):
            rows = vals.get(counter, {}).get(k)
            if not rows:
                return None, None, None
            gmax = max(r[0] for r in rows)
            rows = [r for r in rows if r[0] == gmax]
            return sum(r[2] for r in rows) / len(rows), gmax // rows[0][1], sum(r[3] for r in rows) / len(rows)
        insts, wgs, dur_ns = avg("SQ_INSTS_VALU")
        if insts is None:
            continue
        rotations = wgs * shape["rot_per_wg"]
        gui, _, gui_dur = avg("GRBM_GUI_ACTIVE")
        clock = gui / 8.0 / (gui_dur * 1e-9) if gui else None
        fetch, _, _ = avg("FETCH_SIZE")
        write, _, _ = avg("WRITE_SIZE")
        hit, _, _ = avg("TCC_HIT_sum")
        miss, _, _ = avg("TCC_MISS_sum")
        wave_cycles, _, _ = avg("SQ_WAVE_CYCLES")
        k_out = {
            "rotations_per_launch": rotations,
            "launch_ms": dur_ns * 1e-6,
            "valu_insts_per_launch": insts,
            "valu_insts_per_step_per_wave": insts / (rotations / shape["rot_per_wg"] * shape["waves_per_wg"] * STEPS),
            "clock_hz": clock,
            "valu_pipe_busy": 4.0 * insts / (1024 * clock * dur_ns * 1e-9) if clock else None,
            "waves_per_simd": shape["waves_per_simd"],
            "hbm_bytes_per_launch": (2.0 * fetch + (write or 0.0)) * 1024.0 if fetch is not None else None,
            "fetch_size_kb_raw": fetch, "write_size_kb": write,
            "tcc_hit_rate": hit / (hit + miss) if hit is not None and miss is not None else None,
        }
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "SQ_INSTS_SALU"):
            v, _, _ = avg(c)
            if v is not None:
                k_out[c] = v
        if wave_cycles:
            k_out["SQ_WAVE_CYCLES"] = wave_cycles
        out["kernels"][k] = k_out
    dst = os.path.join(ROOT, "profiles", "kernel_facts.json")
    json.dump(out, open(dst, "w"), indent=1)
    for f in files:      # keep the raw passes beside it, named per round and per pass
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{args.tag}_{os.path.basename(os.path.dirname(f))}_counter_collection.csv"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
