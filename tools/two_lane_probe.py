"""Do a bulk lane and a chain lane share the chip?  (measurement aid for the two-lane scheduler, profiles/r06_two_lane_probe.txt)

A dependent netlist has a narrow critical chain (few gates per step, many sequential steps) beside bulk work nothing waits for.
The level-synchronous scheduler runs them one after the other; the two-lane plan runs the chain's steps on the paired low-latency
kernel on HALF the CUs (cus/2 workgroups of two rotations) while the bulk runs on the batch kernel on the other half (cus/2
workgroups of eight rotations), on two streams.  That only works if the hardware places the two grids side by side: each in-order
stream has at most cus/2 workgroups in flight, so every workgroup finds a free CU whichever CUs the other lane holds.  This script
times the two lanes alone and together."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cufhe_amd as eng  # noqa: E402

api = eng.api


def main():
    eng.SetGPUNum(1)
    rng = np.random.default_rng(1)
    p = api.PARAMS
    bk = rng.integers(0, 2**32, size=p.bk_words, dtype=np.uint64).astype(np.uint32)
    ksk = rng.integers(0, 2**32, size=p.ksk_words, dtype=np.uint64).astype(np.uint32)
    eng.Initialize(bk, ksk)
    cus = api.device_cus()
    W = p.lvl0_words
    bulk_n, chain_n = 8 * (cus // 2), 2 * (cus // 2)
    big = 8 * cus
    a = api.DeviceBuffer(big * W).upload(rng.integers(0, 2**32, size=big * W, dtype=np.uint64).astype(np.uint32))
    b = api.DeviceBuffer(big * W).upload(rng.integers(0, 2**32, size=big * W, dtype=np.uint64).astype(np.uint32))
    o1, o2 = api.DeviceBuffer(big * W), api.DeviceBuffer(big * W)
    sa, sb = api.Stream(0), api.Stream(0)
    sa.Create(); sb.Create()
    bulk_launches, chain_steps = 4, 14

    def bulk(n=bulk_n, shape=1):
        api.set_option("br_shape", shape)
        for _ in range(bulk_launches):
            api.gate_batch(api.NAND, 0, o1, a, b, count=n, stream=sa.st())

    def chain(n=chain_n, shape=2):
        api.set_option("br_shape", shape)
        for _ in range(chain_steps):
            api.gate_batch(api.NAND, 0, o2, a, b, count=n, stream=sb.st())

    def sync(which):
        api._lib.check(api.lib.cufhe_amd_stream_synchronize(0, which.st()))

    def timed(fn_list, label):
        eng.Synchronize()
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            for fn in fn_list:
                fn()
            ends = {}
            # wait for the chain first: its completion time is what a dependent netlist feels
            if chain in [f for f in fn_list] or any(getattr(f, "is_chain", False) for f in fn_list):
                sync(sb); ends["chain_ms"] = (time.perf_counter() - t0) * 1e3
            sync(sa); ends["bulk_ms"] = (time.perf_counter() - t0) * 1e3
            sync(sb); ends["all_ms"] = (time.perf_counter() - t0) * 1e3
            if best is None or ends["all_ms"] < best["all_ms"]:
                best = ends
        print(label, {k: round(v, 2) for k, v in best.items()}, flush=True)
        return best

    print(f"cus {cus}: bulk lane {bulk_launches} x {bulk_n} rotations on the batch kernel ({cus // 2} workgroups), "
          f"chain lane {chain_steps} x {chain_n} on the paired low-latency kernel ({cus // 2} workgroups)", flush=True)
    for _ in range(2):       # warm-up: LDS opt-ins, workspaces, clocks
        bulk(); chain(); eng.Synchronize()
    timed([bulk], "bulk lane alone            ")
    timed([chain], "chain lane alone           ")
    timed([bulk, chain], "both, bulk issued first    ")
    timed([chain, bulk], "both, chain issued first   ")
    # the same chain beside FULL-WIDTH bulk launches (one workgroup per CU): what two streams do without the half-width rule
    wide = lambda: bulk(n=big)
    timed([wide], "full-width bulk alone      ")
    timed([wide, chain], "full-width bulk + chain    ")
    # the chain on the single low-latency kernel (one rotation per workgroup, cus/2 rotations per step) beside half-width bulk
    ll = lambda: chain(n=cus // 2, shape=3)
    ll.is_chain = True
    timed([ll], "ll chain (cus/2 wide) alone")
    timed([bulk, ll], "bulk + ll chain            ")
    api.set_option("br_shape", 0)
    sa.Destroy(); sb.Destroy()
    eng.CleanUp()


if __name__ == "__main__":
    main()
