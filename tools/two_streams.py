"""4096-NAND batches issued on S HIP streams in turn (independent batches, separate outputs): does the key switch of one
batch overlap the blind rotation of the next?   python tools/two_streams.py [streams...]"""
import os
import sys
import time
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
n = int(P.n)
count, steps = 4096, 12
a = rng.integers(0, 2**32, size=(count, n + 1), dtype=np.uint64).astype(np.uint32)
d0 = eng.api.DeviceBuffer(a.size).upload(a)
d1 = eng.api.DeviceBuffer(a.size).upload(a[::-1].copy())
for S in [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]:
    sts = [eng.Stream(0) for _ in range(S)]
    for s in sts:
        s.Create()
    outs = [eng.api.DeviceBuffer(count * (n + 1)) for _ in range(S)]
    for rep in range(2):
        eng.Synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            eng.api.gate_batch(eng.api.NAND, 0, outs[k % S], d0, d1, None, count=count, device=0, stream=sts[k % S].st())
        eng.Synchronize()
        dt = time.perf_counter() - t0
    same = all(np.array_equal(outs[0].download(), o.download()) for o in outs[1:])
    print(f"{S} stream(s): {steps} batches of {count} in {1e3 * dt:.1f} ms = {steps * count / dt / 1e3:.1f} k gates/s, {1e3 * dt / steps:.2f} ms per batch, outputs equal: {same}", flush=True)
    for s in sts:
        s.Destroy()
eng.CleanUp()
