"""Kernel timeline of the two-lane netlist run (rocprofv3 --kernel-trace CSV of tools/bench_api): per lane, the launches of the
last two-lane flush with their durations and the gaps between them.  Measurement aid (profiles/r06_netlist_trace.txt)."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0), int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 1)))
rows.sort()
# the two-lane flushes: blind_rotate_ll2_kernel launches with 128 workgroups (grid 128 x 1024 threads)
ll2 = [r for r in rows if "blind_rotate_ll2" in r[2] and r[3] // max(1, r[4]) <= 128]
if not ll2:
    print("no half-width ll2 launches found")
    sys.exit(0)
# last run of consecutive half-width chain steps
end = ll2[-1][1]
start = end
seg = []
for r in reversed(ll2):
    if start - r[1] > 30_000_000:
        break
    seg.append(r)
    start = r[0]
seg.reverse()
t0 = seg[0][0]
print(f"{len(seg)} chain steps over {(seg[-1][1] - t0) / 1e6:.1f} ms")
print("chain step durations (ms):", " ".join(f"{(r[1] - r[0]) / 1e6:.2f}" for r in seg))
print("chain step periods (start to start, ms):", " ".join(f"{(b[0] - a[0]) / 1e6:.2f}" for a, b in zip(seg, seg[1:])))
inwin = [r for r in rows if r[0] >= t0 - 25_000_000 and r[1] <= seg[-1][1] + 40_000_000]
bulk = [r for r in inwin if r[2].startswith("cufhe_amd::blind_rotate_kernel") or "blind_rotate_kernel(" in r[2]]
print("batch-kernel launches in the window (start ms, duration ms, workgroups):")
for r in bulk:
    print(f"  {(r[0] - t0) / 1e6:8.2f} {(r[1] - r[0]) / 1e6:7.2f} {r[3] // max(1, r[4])}")
ks = [r for r in inwin if "keyswitch" in r[2]]
print("key-switch launches: count", len(ks), "total ms", round(sum(r[1] - r[0] for r in ks) / 1e6, 2))
byname = {}
for r in ks:
    k = r[2].split("(")[0]
    byname.setdefault(k, []).append((r[1] - r[0]) / 1e6)
for k, v in byname.items():
    print(f"  {k}: {len(v)} launches, mean {sum(v) / len(v):.3f} ms")
print("window total ms:", round((max(r[1] for r in inwin) - min(r[0] for r in inwin)) / 1e6, 2))
