"""Gate-batch time at the batch sizes between the launch shapes: python tools/tail_times.py [option=value ...] [counts...]"""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cufhe_amd as eng
rng = np.random.default_rng(1)
P = eng.PARAMS
bk = rng.integers(0, 2**32, size=int(P.bk_words), dtype=np.uint64).astype(np.uint32)
ksk = rng.integers(0, 2**32, size=int(P.ksk_words), dtype=np.uint64).astype(np.uint32)
eng.SetGPUNum(1)
eng.Initialize(bk, ksk)
n = int(P.n)
for kv in [a for a in sys.argv[1:] if "=" in a]:      # library options, e.g. ll2_threshold=2048
    eng.api.set_option(kv.split("=")[0], int(kv.split("=")[1]))
counts = [int(a) for a in sys.argv[1:] if "=" not in a] or [900, 1024, 1025, 1100, 1280, 1281, 1536, 2047, 2048, 3072, 3200, 3328, 3329, 4096]
mx = max(counts)
a = rng.integers(0, 2**32, size=(mx, n + 1), dtype=np.uint64).astype(np.uint32)
d0 = eng.api.DeviceBuffer(a.size).upload(a)
d1 = eng.api.DeviceBuffer(a.size).upload(a[::-1].copy())
dout = eng.api.DeviceBuffer(mx * (n + 1))
for c in counts:
    ts = []
    for _ in range(4):
        eng.Synchronize()
        t0 = time.perf_counter()
        eng.gate_batch(eng.api.NAND, 0, dout, d0, d1, count=c)
        eng.Synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    t = sorted(ts)[1]
    print(f"{c}: {t:.2f} ms  {c / t:.1f} k gates/s", flush=True)
eng.CleanUp()
