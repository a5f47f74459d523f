"""Shared-table key switch by launch size, for the default library or a diagnostic build (what the table pipeline and the digits cost
alone):  python cufhe_amd/build.py --diagnostic=KS_NO_DIGITS; CUFHE_AMD_LIBRARY=cufhe_amd/libcufhe_amd_diag.so python tools/ks_floor.py"""
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
eng.api.set_option("ks_split_threshold", 0); eng.api.set_option("ks_wg_threshold", 0)
print(os.environ.get("CUFHE_AMD_LIBRARY", "default"), " ".join(f"{c}:{t(c):.3f}" for c in (1, 8, 256, 1024, 2048, 4096)), flush=True)
eng.CleanUp()
