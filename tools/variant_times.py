"""Kernel times of one build of the library (CUFHE_AMD_LIBRARY=... selects it): blind rotation launches of the
headline shape, the low-latency shapes, the N = 2048 ring and the parameter-set kernels, HIP events on the launch stream.
   CUFHE_AMD_LIBRARY=cufhe_amd/libcufhe_amd_x.so python tools/variant_times.py [tag]"""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
import cufhe_amd as eng
api = eng.api
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("CUFHE_AMD_LIBRARY", "product"))
skip = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else set()
rng = np.random.default_rng(1)
P = eng.PARAMS
n = int(P.n)
bk = rng.integers(0, 2**32, size=int(P.bk_words), dtype=np.uint64).astype(np.uint32)
ksk = rng.integers(0, 2**32, size=int(P.ksk_words), dtype=np.uint64).astype(np.uint32)
eng.SetGPUNum(1)
eng.Initialize(bk, ksk)
res = {}


def br_ms(f, reps=3):
    f(); eng.Synchronize()
    api.profile_get(reset=True); api.profile_enable(True)
    for _ in range(reps):
        f()
    eng.Synchronize()
    p = api.profile_get(reset=True); api.profile_enable(False)
    return p.blind_rotate_ms / max(p.blind_rotate_launches, 1), p.keyswitch_ms / max(p.keyswitch_launches, 1)


a = rng.integers(0, 2**32, size=(4096, n + 1), dtype=np.uint64).astype(np.uint32)
d0 = api.DeviceBuffer(a.size).upload(a)
d1 = api.DeviceBuffer(a.size).upload(a[::-1].copy())
out = api.DeviceBuffer(a.size)
for count in (4096, 1, 256, 512, 1024):
    if "lvl1" in skip:
        break
    b, k = br_ms(lambda: api.gate_batch(api.NAND, 0, out, d0, d1, count=count))
    res[f"nand{count}"] = (b, k)
    print(f"[{tag}] lvl1 {count:5d} NAND: blind rotate {b:8.3f} ms  key switch {k:6.3f} ms", flush=True)
if "lvl2" not in skip:
    p2 = api.lvl2_params()
    bk2 = rng.integers(0, 2**64, size=int(p2.bk_words), dtype=np.uint64)
    ksk2 = rng.integers(0, 2**32, size=int(p2.ksk_words), dtype=np.uint64).astype(np.uint32)
    api.lvl2_initialize(bk2, ksk2)
    for count in (4096, 1):
        b, k = br_ms(lambda: api.lvl2_gate_batch(api.NAND, out, d0, d1, None, count=count), reps=2)
        print(f"[{tag}] lvl2 {count:5d} NAND: blind rotate {b:8.3f} ms  key switch {k:6.3f} ms", flush=True)
if "ps" not in skip:
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
        for count in (4096, 1):
            ts = []
            for _ in range(3):
                eng.Synchronize()
                t0 = time.perf_counter()
                api.ps_gate_batch(ps, api.NAND, eo, e0, e1, count=count)
                eng.Synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
            print(f"[{tag}] ps {p.name.decode():8s} {count:5d} NAND: {sorted(ts)[1]:8.3f} ms (gate batch, wall)", flush=True)
eng.CleanUp()
