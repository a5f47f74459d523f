"""Gate-batch time (blind rotation + key switch, host clock around enqueue .. Synchronize) for the launch sizes the two
low-latency kernels serve, default launch rules: python tools/ll_times.py [count ...]"""
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
counts = [int(a) for a in sys.argv[1:]] or [1, 64, 256, 320, 512, 768, 1024]
mx = max(counts)
a = rng.integers(0, 2**32, size=(mx, n + 1), dtype=np.uint64).astype(np.uint32)
d0 = eng.api.DeviceBuffer(a.size).upload(a)
d1 = eng.api.DeviceBuffer(a.size).upload(a[::-1].copy())
dout = eng.api.DeviceBuffer(mx * (n + 1))
for c in counts:
    ts = []
    for _ in range(9):
        eng.Synchronize()
        t0 = time.perf_counter()
        eng.gate_batch(eng.api.NAND, 0, dout, d0, d1, count=c)
        eng.Synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    print(f"{c:5d} gates: median {ts[4]:.3f} ms  min {ts[0]:.3f} ms", flush=True)
eng.CleanUp()
