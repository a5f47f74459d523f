import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cufhe_amd as eng
rng = np.random.default_rng(1)
P = eng.PARAMS
bk = rng.integers(0, 2**32, size=int(P.bk_words), dtype=np.uint64).astype(np.uint32)
ksk = rng.integers(0, 2**32, size=int(P.ksk_words), dtype=np.uint64).astype(np.uint32)
eng.SetGPUNum(1); eng.Initialize(bk, ksk)
n = int(P.n)
mx = 2048
a = rng.integers(0, 2**32, size=(mx, n+1), dtype=np.uint64).astype(np.uint32)
d0 = eng.api.DeviceBuffer(a.size).upload(a); d1 = eng.api.DeviceBuffer(a.size).upload(a[::-1].copy())
dout = eng.api.DeviceBuffer(mx*(n+1))
def t(count, reps=5):
    ts=[]
    for _ in range(reps):
        eng.Synchronize(); t0=time.perf_counter()
        eng.gate_batch(eng.api.NAND, 0, dout, d0, d1, count=count)
        eng.Synchronize(); ts.append(1e3*(time.perf_counter()-t0))
    return sorted(ts)[len(ts)//2]
for name,(ll,wg) in dict(ll=(1<<30,1<<30), wg=(0,1<<30), batch=(0,0)).items():
    eng.api.set_option("ll_threshold", ll); eng.api.set_option("wg_threshold", wg)
    print(name, " ".join(f"{c}:{t(c):.2f}" for c in (1, 16, 64, 256, 512, 1024, 2048)), flush=True)
eng.CleanUp()
