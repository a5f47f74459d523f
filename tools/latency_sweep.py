"""Gate-batch time by batch size and kernel choice (1 GPU): the numbers behind the launch-shape
thresholds of cufhe_amd/csrc/capi.hip (launch_blind_rotate).  python tools/latency_sweep.py"""
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
mx = 8192
a = rng.integers(0, 2**32, size=(mx, n + 1), dtype=np.uint64).astype(np.uint32)
d0 = eng.api.DeviceBuffer(a.size).upload(a)
d1 = eng.api.DeviceBuffer(a.size).upload(a[::-1].copy())
dout = eng.api.DeviceBuffer(mx * (n + 1))


def t(count, reps=5):
    ts = []
    for _ in range(reps):
        eng.Synchronize()
        t0 = time.perf_counter()
        eng.gate_batch(eng.api.NAND, 0, dout, d0, d1, count=count)
        eng.Synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[len(ts) // 2]


def opts(ll, ll2, half, split):
    eng.api.set_option("ll_threshold", ll)
    eng.api.set_option("ll2_threshold", ll2)
    eng.api.set_option("half_threshold", half)
    eng.api.set_option("tail_split", split)


BIG = 1 << 30
small = (1, 16, 64, 256, 384, 512, 640, 768, 1024, 1280, 1536, 2048)
for name, o in dict(ll=(BIG, 0, 0, 0), ll2=(0, BIG, 0, 0), half=(0, 0, BIG, 0), batch=(0, 0, 0, 0)).items():
    opts(*o)
    print(name, " ".join(f"{c}:{t(c):.2f}" for c in small), flush=True)
big = (2049, 2112, 2304, 2560, 2816, 3072, 3500, 4096, 4097, 4352, 5000, 6144, 8192)
opts(1280, 0, 0, 0)
print("without tail split     ", " ".join(f"{c}:{t(c, 3):.2f}" for c in big), flush=True)
opts(-1, -1, -1, 1)
print("tail split (defaults)  ", " ".join(f"{c}:{t(c, 3):.2f}" for c in big), flush=True)
print("gates/s (defaults)     ", " ".join(f"{c}:{c / t(c, 3) * 1e3:.0f}" for c in big), flush=True)
eng.CleanUp()
