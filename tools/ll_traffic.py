"""Launches of the low-latency kernel at one batch size, for rocprofv3 counter passes (L2 hit rate and HBM
bytes of a launch that fills the chip with one rotation per CU):
   rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d out -o pmc --output-format csv -- python3 tools/ll_traffic.py 256"""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cufhe_amd as eng
count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(1)
P = eng.PARAMS
bk = rng.integers(0, 2**32, size=int(P.bk_words), dtype=np.uint64).astype(np.uint32)
ksk = rng.integers(0, 2**32, size=int(P.ksk_words), dtype=np.uint64).astype(np.uint32)
eng.SetGPUNum(1)
eng.Initialize(bk, ksk)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    eng.api.set_option(k, int(v))
n = int(P.n)
a = rng.integers(0, 2**32, size=(count, n + 1), dtype=np.uint64).astype(np.uint32)
d0 = eng.api.DeviceBuffer(a.size).upload(a)
d1 = eng.api.DeviceBuffer(a.size).upload(a[::-1].copy())
dout = eng.api.DeviceBuffer(count * (n + 1))
eng.api.set_option("ll_threshold", 1 << 30)
ts = []
for _ in range(reps):
    eng.Synchronize()
    t0 = time.perf_counter()
    eng.gate_batch(eng.api.NAND, 0, dout, d0, d1, count=count)
    eng.Synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print(f"low-latency kernel, {count} gates: " + " ".join(f"{t:.2f}" for t in ts) + " ms", flush=True)
eng.CleanUp()
