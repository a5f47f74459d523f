"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (per stream the trace has no gaps: this is the
device timeline as a whole).  usage: python tools/kernel_gaps.py <..._kernel_trace.csv> [min_gap_us]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
end = None
tot_gap = tot_busy = 0
names = {}
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end is not None and s > end:
        g = (s - end) / 1e3
        if g < 2000:      # gaps above 2 ms are between phases of the benchmark, not between launches
            tot_gap += g
            key = (prev.split("(")[0][-40:], r["Kernel_Name"].split("(")[0][-40:])
            a = names.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += g
    tot_busy += (e - s) / 1e3
    end = max(end or 0, e)
    prev = r["Kernel_Name"]
print(f"kernels {len(rows)}  busy {tot_busy / 1e3:.2f} ms  idle between launches {tot_gap / 1e3:.2f} ms")
for k, (n, g) in sorted(names.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {n:5d} x {g / n:8.1f} us   {k[0]} -> {k[1]}")
