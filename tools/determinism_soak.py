"""Repeat the same launches and require bit-identical outputs every time (a race in a staging protocol -- LDS-DMA pieces, row
barriers, the scheduler's buffers -- shows as a rare word mismatch):   python tools/determinism_soak.py [reps]"""
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cufhe_amd as eng
api = eng.api
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(7)
P = eng.PARAMS
n = int(P.n)
bk = rng.integers(0, 2**32, size=int(P.bk_words), dtype=np.uint64).astype(np.uint32)
ksk = rng.integers(0, 2**32, size=int(P.ksk_words), dtype=np.uint64).astype(np.uint32)
eng.SetGPUNum(1)
eng.Initialize(bk, ksk)
bad = 0


def soak(name, launch, out, count, reps):
    global bad
    launch(); eng.Synchronize()
    ref = out.download().copy()
    mism = 0
    for _ in range(reps):
        launch(); eng.Synchronize()
        if not np.array_equal(out.download(), ref):
            mism += 1
    bad += mism
    print(f"{name:28s} {count:5d} per launch x {reps:4d} launches: {mism} differing", flush=True)


a = rng.integers(0, 2**32, size=(4096, n + 1), dtype=np.uint64).astype(np.uint32)
d0 = api.DeviceBuffer(a.size).upload(a)
d1 = api.DeviceBuffer(a.size).upload(a[::-1].copy())
d2 = api.DeviceBuffer(a.size).upload(np.roll(a, 7, axis=0).copy())
out = api.DeviceBuffer(a.size)
for count in (4096, 2049, 512, 256, 1):
    soak("NAND", lambda: api.gate_batch(api.NAND, 0, out, d0, d1, count=count), out, count, reps if count > 600 else 2 * reps)
soak("MUX", lambda: api.gate_batch(api.MUX, 0, out, d0, d1, d2, count=4096), out, 4096, reps // 2)
p2 = api.lvl2_params()
bk2 = rng.integers(0, 2**64, size=int(p2.bk_words), dtype=np.uint64)
ksk2 = rng.integers(0, 2**32, size=int(p2.ksk_words), dtype=np.uint64).astype(np.uint32)
api.lvl2_initialize(bk2, ksk2)
soak("NAND through the N=2048 ring", lambda: api.lvl2_gate_batch(api.NAND, out, d0, d1, None, count=4096), out, 4096, max(reps // 10, 3))
for ps in range(api.ps_count()):
    p = api.ps_params(ps)
    pbk = rng.integers(0, 2**32, size=int(p.bk_words), dtype=np.uint64).astype(np.uint32)
    pksk = rng.integers(0, 2**32, size=int(p.ksk_words), dtype=np.uint64).astype(np.uint32)
    api.ps_initialize(ps, pbk, pksk)
    w = int(p.lvl0_words)
    x = rng.integers(0, 2**32, size=(4096, w), dtype=np.uint64).astype(np.uint32)
    e0 = api.DeviceBuffer(x.size).upload(x)
    e1 = api.DeviceBuffer(x.size).upload(x[::-1].copy())
    eo = api.DeviceBuffer(x.size)
    soak("parameter set " + p.name.decode(), lambda: api.ps_gate_batch(ps, api.NAND, eo, e0, e1, count=4096), eo, 4096, reps // 4)
    for count in (700, 256, 1):      # the workgroup-per-rotation kernel (key polynomials read a step ahead, LDS sums)
        soak("parameter set " + p.name.decode(), lambda: api.ps_gate_batch(ps, api.NAND, eo, e0, e1, count=count), eo, count, reps)
eng.CleanUp()
print("TOTAL differing launches:", bad)
sys.exit(1 if bad else 0)
