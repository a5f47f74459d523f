"""Cycles per phase of blind_rotate_lvl2_kernel (diagnostic build with CUFHE_AMD_ABL_PHASES):
   CUFHE_AMD_LIBRARY=cufhe_amd/libcufhe_amd_diag.so python tools/lvl2_phases.py"""
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cufhe_amd as eng
rng = np.random.default_rng(1)
p2 = eng.api.lvl2_params()
bk = rng.integers(0, 2**64, size=int(p2.bk_words), dtype=np.uint64)
ksk = rng.integers(0, 2**32, size=int(p2.ksk_words), dtype=np.uint64).astype(np.uint32)
eng.SetGPUNum(1)
eng.api.lvl2_initialize(bk, ksk)
n, N = int(p2.n), int(p2.N)
kernel = int(os.environ.get("LVL2_KERNEL", "1"))       # 1: four quarter waves (kernels_lvl2q.hip.h), 0: eight half waves
eng.api.set_option("lvl2_kernel", kernel)
if kernel == 1:
    names = ["barrier", "rot reads+barrier", "digits+barrier", "split+fwd x8", "products x8", "barrier", "inverse x6", "exch barrier x6", "last stages+lift x6", "publish"]
    waves = 4
else:
    names = ["barrier", "rot reads+barrier", "digits+barrier", "fwd h0", "prod h0", "fwd h1", "prod h1", "barrier", "inverse jobs", "barrier", "-", "-", "recombine"]
    waves = 8
for count in (1, 2, 4096):
    tl = rng.integers(0, 2**32, size=(count, n + 1), dtype=np.uint64).astype(np.uint32)
    d = eng.api.DeviceBuffer(tl.size).upload(tl)
    acc = eng.api.DeviceBuffer(count * 2 * N * 2)
    eng.api.lvl2_blind_rotate_batch(d, acc, count)
    eng.api.lvl2_blind_rotate_batch(d, acc, count)
    eng.Synchronize()
    w = acc.download()[: 2 * N * 2].view(np.uint64)
    print(f"--- {count} rotation(s): cycles per step (630 steps), workgroup 0, by wave")
    for wave in range(waves):
        c = w[2048 + wave * 16: 2048 + wave * 16 + len(names)] / 630.0
        print(f"wave {wave}: " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, c)) + f"   total {c.sum():7.0f}")
eng.CleanUp()
