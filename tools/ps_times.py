"""Throughput of the parameter-set kernels (kernels_ps.hip.h): 4096 NAND per launch on random keys.
   python tools/ps_times.py"""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cufhe_amd as eng
rng = np.random.default_rng(1)
eng.SetGPUNum(1)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for ps in range(eng.api.ps_count()):
    p = eng.api.ps_params(ps)
    bk = rng.integers(0, 2**32, size=int(p.bk_words), dtype=np.uint64).astype(np.uint32)
    ksk = rng.integers(0, 2**32, size=int(p.ksk_words), dtype=np.uint64).astype(np.uint32)
    eng.api.ps_initialize(ps, bk, ksk)
    w = int(p.lvl0_words)
    a = rng.integers(0, 2**32, size=(count, w), dtype=np.uint64).astype(np.uint32)
    d0 = eng.api.DeviceBuffer(a.size).upload(a)
    d1 = eng.api.DeviceBuffer(a.size).upload(a[::-1].copy())
    dout = eng.api.DeviceBuffer(count * w)
    ts = []
    for _ in range(4):
        eng.Synchronize()
        t0 = time.perf_counter()
        eng.api.ps_gate_batch(ps, eng.api.NAND, dout, d0, d1, count=count)
        eng.Synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    t = sorted(ts)[1]
    print(f"{p.name.decode():10s} n={p.n} N={p.N} k={p.k} l={p.l} Bgbit={p.Bgbit} limbs={p.key_limbs}: {count} NAND {t:.2f} ms = {count / t:.1f} k gates/s", flush=True)
eng.CleanUp()
