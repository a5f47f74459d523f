"""Cycles per phase of the low-latency blind-rotate kernel (diagnostic build with CUFHE_AMD_ABL_PHASES):
   CUFHE_AMD_LIBRARY=cufhe_amd/libcufhe_amd_diag.so python tools/ll_phases.py"""
import os
import sys
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
n, N = int(P.n), int(P.N)
for count in (1, 256):
    tl = rng.integers(0, 2**32, size=(count, n + 1), dtype=np.uint64).astype(np.uint32)
    d = eng.api.DeviceBuffer(tl.size).upload(tl)
    acc = eng.api.DeviceBuffer(count * 2 * N)
    eng.api.set_option("ll_threshold", 1 << 30)
    eng.api.blind_rotate_batch(d, acc, count)
    eng.api.blind_rotate_batch(d, acc, count)
    eng.Synchronize()
    w = acc.download()[: 2 * N].view(np.uint64)
    names = ["row work", "barrier 1", "inverse work", "barrier 2", "final stage", "barrier 3", "decompose", "barrier 4"]
    print(f"--- {count} rotation(s): cycles per step (630 steps), by wave")
    for wave in (0, 3, 4, 7, 8, 11, 12, 15):
        c = w[16 + wave * 8: 16 + wave * 8 + 8] / 630.0
        print(f"wave {wave:2d}: " + "  ".join(f"{nm} {v:7.0f}" for nm, v in zip(names, c)) + f"   total {c.sum():7.0f}")
eng.CleanUp()
