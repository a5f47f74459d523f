"""The dependent netlist, again and again, level by level and gate by gate on two lanes: every repetition must produce the words of the
first level-by-level run (a missing edge between the lanes, a result fetched too late, a buffer recycled too early shows as a word
mismatch now and then).   python tools/two_lane_soak.py [reps] [adders] [bits]        (profiles/r06_two_lane_soak.txt)"""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
import numpy as np
import cufhe_amd as eng

api = eng.api
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
A = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rng = np.random.default_rng(11)
P = eng.PARAMS
eng.SetGPUNum(1)
eng.Initialize(rng.integers(0, 2**32, size=int(P.bk_words), dtype=np.uint64).astype(np.uint32),
               rng.integers(0, 2**32, size=int(P.ksk_words), dtype=np.uint64).astype(np.uint32))
W = int(P.lvl0_words)
x = [api.Ctxt(0) for _ in range(A * B)]
y = [api.Ctxt(0) for _ in range(A * B)]
s = [api.Ctxt(0) for _ in range(A * B)]
carry = [api.Ctxt(0) for _ in range(A)]
t1 = [api.Ctxt(0) for _ in range(A)]
t2 = [api.Ctxt(0) for _ in range(A)]
xin = rng.integers(0, 2**32, size=(A * B, W), dtype=np.uint64).astype(np.uint32)
yin = rng.integers(0, 2**32, size=(A * B, W), dtype=np.uint64).astype(np.uint32)
cin = rng.integers(0, 2**32, size=(A, W), dtype=np.uint64).astype(np.uint32)
sts = [api.Stream() for _ in range(64)]
for st in sts:
    st.Create()


def run():
    for i in range(A * B):
        x[i].tlwehost[:] = xin[i]
        y[i].tlwehost[:] = yin[i]
    for i in range(A):
        carry[i].tlwehost[:] = cin[i]
    api.sched_stats(reset=True)
    t0 = time.perf_counter()
    for i in range(A):
        st = sts[i % len(sts)]
        for k in range(B):
            X, Y, S, C = x[i * B + k], y[i * B + k], s[i * B + k], carry[i]
            api.Xor(t1[i], X, Y, st)
            api.Xor(S, t1[i], C, st)
            api.And(t2[i], t1[i], C, st)
            api.And(t1[i], X, Y, st)
            api.Or(C, t1[i], t2[i], st)
    api.Synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    out = np.stack([c.tlwehost.copy() for c in s + carry + t1 + t2])
    return out, ms, api.sched_stats()


bad = 0
api.set_option("sched_two_lane", 0)
ref, ms, st = run()
print(f"{A} x {B}-bit adders, {5 * A * B} gates; reference run level by level: {ms:.1f} ms (python issue time included), {st.launch_sequences} launch sequences", flush=True)
for mode, name in ((2, "two lanes (forced)"), (1, "two lanes by the cost model"), (0, "level by level")):
    api.set_option("sched_two_lane", mode)
    mism, flushes, best = 0, 0, 1e30
    for _ in range(reps):
        out, ms, st = run()
        mism += not np.array_equal(out, ref)
        flushes += st.two_lane_groups
        best = min(best, ms)
    bad += mism
    print(f"{name:30s}: {reps} repetitions, {mism} with a word different from the reference run, {flushes} flushes on two lanes, best {best:.1f} ms", flush=True)
api.set_option("sched_two_lane", 1)
for st in sts:
    st.Destroy()
eng.CleanUp()
print("SOAK", "FAILED" if bad else "PASSED")
sys.exit(1 if bad else 0)
