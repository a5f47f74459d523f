"""A/B of the two N = 2048 blind-rotate kernels: ms per launch of `count` rotations (HIP events on the launch stream),
   python tools/lvl2_ab.py [count]     (CUFHE_AMD_LIBRARY=... for a diagnostic build)"""
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cufhe_amd as eng
count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(1)
p2 = eng.api.lvl2_params()
bk = rng.integers(0, 2**64, size=int(p2.bk_words), dtype=np.uint64)
ksk = rng.integers(0, 2**32, size=int(p2.ksk_words), dtype=np.uint64).astype(np.uint32)
eng.SetGPUNum(1)
eng.api.lvl2_initialize(bk, ksk)
n, N = int(p2.n), int(p2.N)
tl = rng.integers(0, 2**32, size=(count, n + 1), dtype=np.uint64).astype(np.uint32)
d = eng.api.DeviceBuffer(tl.size).upload(tl)
acc = eng.api.DeviceBuffer(count * 2 * N * 2)
ref = None
for name, k in (("eight half waves (kernels_lvl2.hip.h)", 0), ("four quarter waves (kernels_lvl2q.hip.h)", 1), ("eight half waves", 0), ("four quarter waves", 1)):
    eng.api.set_option("lvl2_kernel", k)
    eng.api.lvl2_blind_rotate_batch(d, acc, count)
    eng.Synchronize()
    eng.api.profile_get(reset=True)
    eng.api.profile_enable(True)
    for _ in range(2):
        eng.api.lvl2_blind_rotate_batch(d, acc, count)
    eng.Synchronize()
    pr = eng.api.profile_get(reset=True)
    eng.api.profile_enable(False)
    ms = pr.blind_rotate_ms / max(pr.blind_rotate_launches, 1)
    w = acc.download().copy()
    same = "" if ref is None else ("  words == first kernel: %s" % bool(np.array_equal(w, ref)))
    if ref is None:
        ref = w
    print(f"{name:45s} {ms:8.2f} ms per {count} rotations   sweep fraction {count * 165150720 / (ms * 1e-3) / 8e12:.3f}{same}", flush=True)
eng.CleanUp()
