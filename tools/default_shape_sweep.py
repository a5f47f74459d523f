"""Gate-batch time of the DEFAULT launch-shape rules by batch size (where the started-round steps are):
   python tools/default_shape_sweep.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa
import cufhe_amd as eng
api = eng.api
rng = np.random.default_rng(1)
P = eng.PARAMS; n = int(P.n)
bk = rng.integers(0, 2**32, size=int(P.bk_words), dtype=np.uint64).astype(np.uint32)
ksk = rng.integers(0, 2**32, size=int(P.ksk_words), dtype=np.uint64).astype(np.uint32)
eng.SetGPUNum(1); eng.Initialize(bk, ksk)
a = rng.integers(0, 2**32, size=(4096, n + 1), dtype=np.uint64).astype(np.uint32)
d0 = api.DeviceBuffer(a.size).upload(a); d1 = api.DeviceBuffer(a.size).upload(a[::-1].copy()); out = api.DeviceBuffer(a.size)
for c in (1, 64, 128, 256, 257, 320, 384, 448, 512, 513, 640, 768, 769, 896, 1024, 1025, 1152, 1280, 1408, 1536, 1537, 1600, 1792, 1920, 2047, 2048):
    ts = []
    for _ in range(4):
        eng.Synchronize(); t0 = time.perf_counter()
        api.gate_batch(api.NAND, 0, out, d0, d1, count=c); eng.Synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    t = sorted(ts)[1]
    print(f"{c:5d} gates: {t:7.3f} ms  {c / t:7.1f} k gates/s", flush=True)
eng.CleanUp()
