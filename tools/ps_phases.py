"""Cycles per phase of the parameter-set workgroup-per-rotation kernel (diagnostic build with CUFHE_AMD_ABL_PHASES):
   python cufhe_amd/build.py --diagnostic=PHASES && CUFHE_AMD_LIBRARY=cufhe_amd/libcufhe_amd_diag.so python tools/ps_phases.py"""
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cufhe_amd as eng
rng = np.random.default_rng(1)
eng.SetGPUNum(1)
api = eng.api
api.set_option("ps_batch_threshold", 1 << 30)
names = ["decompose", "forward", "products", "barrier 1", "read sums", "inverse", "lift + acc", "barrier 2"]
for ps in range(api.ps_count()):
    p = api.ps_params(ps)
    bk = rng.integers(0, 2**32, size=int(p.bk_words), dtype=np.uint64).astype(np.uint32)
    ksk = rng.integers(0, 2**32, size=int(p.ksk_words), dtype=np.uint64).astype(np.uint32)
    api.ps_initialize(ps, bk, ksk)
    w, tw = int(p.lvl0_words), (int(p.k) + 1) * int(p.N)
    rows = (int(p.k) + 1) * int(p.l)
    for count in (1, 256):
        tl = rng.integers(0, 2**32, size=(count, w), dtype=np.uint64).astype(np.uint32)
        d = api.DeviceBuffer(tl.size).upload(tl)
        acc = api.DeviceBuffer(count * tw)
        api.ps_blind_rotate_batch(ps, d, acc, count, -1)
        api.ps_blind_rotate_batch(ps, d, acc, count, -1)
        eng.Synchronize()
        v = acc.download()[:tw].view(np.uint64)
        print(f"--- {p.name.decode()}, {count} rotation(s): cycles per step ({p.n} steps), by wave (rows on waves 0..{rows - 1}, sums on the first waves)")
        for wave in sorted({0, 1, rows - 1, min(rows, 7)}):
            c = v[16 + wave * 8: 16 + wave * 8 + 8] / float(p.n)
            print(f"wave {wave:2d}: " + "  ".join(f"{nm} {x:6.0f}" for nm, x in zip(names, c)) + f"   total {c.sum():7.0f}")
eng.CleanUp()
