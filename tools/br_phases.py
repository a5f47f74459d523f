"""Row-barrier wait of blind_rotate_kernel per wave (diagnostic build with CUFHE_AMD_ABL_PHASES):
   CUFHE_AMD_LIBRARY=cufhe_amd/libcufhe_amd_diag.so python tools/br_phases.py"""
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
eng.api.set_option("ll_threshold", 0)
eng.api.set_option("half_threshold", 0)
for count in (2048, 4096):
    tl = rng.integers(0, 2**32, size=(count, n + 1), dtype=np.uint64).astype(np.uint32)
    d = eng.api.DeviceBuffer(tl.size).upload(tl)
    acc = eng.api.DeviceBuffer(count * 2 * N)
    eng.api.blind_rotate_batch(d, acc, count)
    eng.api.blind_rotate_batch(d, acc, count)
    eng.Synchronize()
    w = acc.download().reshape(count, 2 * N)[:, :4].copy().view(np.uint64)     # [rotation][total, wait]
    tot, wait = w[:, 0].astype(np.float64) / 630, w[:, 1].astype(np.float64) / 630
    print(f"--- {count} rotations: cycles per CMux step; wave position in workgroup -> total / in row barriers (6 per step)")
    for wv in range(8):
        sel = np.arange(count) % 8 == wv
        print(f"wave {wv}: total {tot[sel].mean():8.0f}  barrier wait {wait[sel].mean():7.0f} ({100 * wait[sel].mean() / tot[sel].mean():4.1f} %)"
              f"   min/max total over workgroups {tot[sel].min():8.0f} / {tot[sel].max():8.0f}")
eng.CleanUp()
