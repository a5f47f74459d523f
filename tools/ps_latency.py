"""Small launches of the parameter-set path (workgroup-per-rotation kernel): python tools/ps_latency.py"""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cufhe_amd as eng
rng = np.random.default_rng(1)
eng.SetGPUNum(1)
for ps in range(eng.api.ps_count()):
    p = eng.api.ps_params(ps)
    bk = rng.integers(0, 2**32, size=int(p.bk_words), dtype=np.uint64).astype(np.uint32)
    ksk = rng.integers(0, 2**32, size=int(p.ksk_words), dtype=np.uint64).astype(np.uint32)
    eng.api.ps_initialize(ps, bk, ksk)
    w = int(p.lvl0_words)
    out = []
    for count in (1, 64, 256, 512):
        a = rng.integers(0, 2**32, size=(count, w), dtype=np.uint64).astype(np.uint32)
        d0 = eng.api.DeviceBuffer(a.size).upload(a)
        d1 = eng.api.DeviceBuffer(a.size).upload(a[::-1].copy())
        dout = eng.api.DeviceBuffer(count * w)
        ts = []
        for _ in range(5):
            eng.Synchronize()
            t0 = time.perf_counter()
            eng.api.ps_gate_batch(ps, eng.api.NAND, dout, d0, d1, count=count)
            eng.Synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        out.append(f"{count}:{sorted(ts)[2]:.2f}")
    print(f"{p.name.decode():10s} ms per launch  " + "  ".join(out), flush=True)
eng.CleanUp()
