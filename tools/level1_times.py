"""Gate-batch time on LEVEL-1 ciphertexts (key switch first, then blind rotate: Nand<lvl1param>) and of
the TRLWE-level batch entry points, 4096 per launch.  python tools/level1_times.py"""
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
n, N = int(P.n), int(P.N)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def timeit(f, reps=4):
    ts = []
    for _ in range(reps):
        eng.Synchronize()
        t0 = time.perf_counter()
        f()
        eng.Synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[1]


for level, w in ((0, n + 1), (1, N + 1)):
    a = rng.integers(0, 2**32, size=(count, w), dtype=np.uint64).astype(np.uint32)
    d = [eng.api.DeviceBuffer(a.size).upload(a) for _ in range(3)]
    out = eng.api.DeviceBuffer(count * w)
    for name, op in (("NAND", eng.api.NAND), ("MUX", eng.api.MUX), ("NOT", eng.api.NOT)):
        t = timeit(lambda: eng.gate_batch(op, level, out, d[0], d[1], d[2], count=count))
        print(f"level {level} {name}: {count} gates {t:.2f} ms = {count / t:.1f} k gates/s", flush=True)
eng.CleanUp()
