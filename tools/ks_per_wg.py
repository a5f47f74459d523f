"""Shared-table key switch by ciphertexts per workgroup (cufhe_amd_set_option("ks_per_wg", n)):  python tools/ks_per_wg.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa
import cufhe_amd as eng
rng = np.random.default_rng(1)
P = eng.PARAMS
bk = rng.integers(0, 2**32, size=int(P.bk_words), dtype=np.uint64).astype(np.uint32)
ksk = rng.integers(0, 2**32, size=int(P.ksk_words), dtype=np.uint64).astype(np.uint32)
eng.SetGPUNum(1); eng.Initialize(bk, ksk)
n, N = int(P.n), int(P.N)
a = rng.integers(0, 2**32, size=(4096, N + 1), dtype=np.uint64).astype(np.uint32)
d1 = eng.api.DeviceBuffer(a.size).upload(a); d0 = eng.api.DeviceBuffer(4096 * (n + 1))
def t(count, reps=9):
    ts = []
    for _ in range(reps):
        eng.Synchronize(); t0 = time.perf_counter(); eng.api.keyswitch_batch(d1, d0, count); eng.Synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[len(ts) // 2]
eng.api.set_option("ks_split_threshold", 0); eng.api.set_option("ks_wg_threshold", 0)
ref = {}
for per in (16, 12, 8, 6, 4, 2, -1):
    eng.api.set_option("ks_per_wg", per)
    print(f"{per:3d}", " ".join(f"{c}:{t(c):.3f}" for c in (256, 512, 1024, 1536, 2048, 3072, 4096)), flush=True)
    ref[per] = d0.download().copy()
print("same words:", all(bool(np.array_equal(ref[16], v)) for v in ref.values()))
eng.api.set_option("ks_split_threshold", -1); eng.api.set_option("ks_wg_threshold", -1)
print("defaults", " ".join(f"{c}:{t(c):.3f}" for c in (256, 512, 1024, 1280, 1536, 1792, 2048, 3072, 4096)), flush=True)
eng.CleanUp()
