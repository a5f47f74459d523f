"""Shared-table key switch by (ciphertexts per workgroup, runs of j) against the other kernels, by launch size:  python tools/ks_slices.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cufhe_amd as eng
rng = np.random.default_rng(1)
P = eng.PARAMS
bk = rng.integers(0, 2**32, size=int(P.bk_words), dtype=np.uint64).astype(np.uint32)
ksk = rng.integers(0, 2**32, size=int(P.ksk_words), dtype=np.uint64).astype(np.uint32)
eng.SetGPUNum(1); eng.Initialize(bk, ksk)
n, N = int(P.n), int(P.N)
mx = 4096
a = rng.integers(0, 2**32, size=(mx, N + 1), dtype=np.uint64).astype(np.uint32)
d1 = eng.api.DeviceBuffer(a.size).upload(a)
d0 = eng.api.DeviceBuffer(mx * (n + 1))


def t(count, reps=9):
    ts = []
    for _ in range(reps):
        eng.Synchronize(); t0 = time.perf_counter(); eng.api.keyswitch_batch(d1, d0, count); eng.Synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[len(ts) // 2]


counts = (1, 8, 16, 32, 64, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096)
BIG = 1 << 30
for name, o in (("split", (BIG, BIG)), ("wg", (0, BIG)), ("shared", (0, 0)), ("defaults", (-1, -1))):
    eng.api.set_option("ks_split_threshold", o[0]); eng.api.set_option("ks_wg_threshold", o[1])
    print(f"{name:9s}", " ".join(f"{c}:{t(c):.3f}" for c in counts), flush=True)
eng.api.set_option("ks_split_threshold", 0); eng.api.set_option("ks_wg_threshold", 0)
for per, sl in ((16, 1), (16, 2), (16, 4), (16, 8), (16, 16), (16, 32), (16, 64), (8, 2), (8, 4)):
    eng.api.set_option("ks_per_wg", per); eng.api.set_option("ks_slices", sl)
    print(f"{per:2d} x {sl:2d}  ", " ".join(f"{c}:{t(c):.3f}" for c in counts if (c + per - 1) // per * sl <= 1024), flush=True)
eng.CleanUp()
