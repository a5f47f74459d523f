"""The N>1 path of bench.py on CPU: world_size-2 gloo ranks (no GPU).

Gates shard with no data-path collective: each rank evaluates its contiguous
shard (here with the CPU oracle standing in for the rank's GPU), ranks meet only in a barrier
and a MAX reduction of the elapsed time, and the union of the shards equals the single-rank
result word for word."""
import os
import socket
import subprocess
import sys

import numpy as np

import oracle_lib as ol

ROOT = ol.ROOT

WORKER = r'''
import os, sys, time, importlib.util
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("d", os.path.join(ROOT, "cufhe_amd", "dist.py"))
d = importlib.util.module_from_spec(spec); spec.loader.exec_module(d)
rank, local_rank, world = d.rank_env()
assert d.device_for_rank(local_rank, 8) == local_rank and d.device_for_rank(local_rank, 1) == 0
d.pin_gpu()
assert os.environ["HIP_VISIBLE_DEVICES"] == str(local_rank)
import numpy as np, torch, torch.distributed as dist
import oracle_lib as ol
dist.init_process_group("gloo", rank=rank, world_size=world)
L = ol.load(); keys = ol.Keys(L, seed=1)
count = 12
bits = np.random.default_rng(5).integers(0, 2, size=(2, count)).astype(np.uint8)
ins = [keys.encrypt(bits[i], 0, seed=300 + i) for i in range(2)]
ops = np.array([[3, 4, 5, 0][g % 4] for g in range(count)], np.int32)     # AND OR XOR NAND (config 3)
lo, hi = d.shard(count, rank, world)
dist.barrier(); t0 = time.perf_counter()
out = keys.gate_batch(ops[lo:hi], 0, ins[0][lo:hi], ins[1][lo:hi])
dist.barrier(); el = d.max_over_ranks(time.perf_counter() - t0 + rank, dist)
assert el >= world - 1                      # MAX over ranks, not this rank's own time
np.save(os.path.join(OUT, f"shard{rank}.npy"), out)
np.save(os.path.join(OUT, f"range{rank}.npy"), np.array([lo, hi]))
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_function():
    import importlib.util
    spec = importlib.util.spec_from_file_location("d", os.path.join(ROOT, "cufhe_amd", "dist.py"))
    d = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(d)
    for count in (0, 1, 7, 4096, 32768):
        for world in (1, 2, 3, 4, 8):
            spans = [d.shard(count, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == count
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= -(-count // world)
    assert d.visible_device_for(3, "4,5,6,7") == "7" and d.visible_device_for(2, None) == "2"


def test_two_ranks_gloo(tmp_path, keys):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="3")
        env.pop("HIP_VISIBLE_DEVICES", None)
        code = f"ROOT={ROOT!r}\nOUT={str(tmp_path)!r}\n" + WORKER
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    count = 12
    bits = np.random.default_rng(5).integers(0, 2, size=(2, count)).astype(np.uint8)
    ins = [keys.encrypt(bits[i], 0, seed=300 + i) for i in range(2)]
    ops = np.array([[3, 4, 5, 0][g % 4] for g in range(count)], np.int32)
    want = keys.gate_batch(ops, 0, ins[0], ins[1])
    got = np.zeros_like(want)
    for rank in range(2):
        lo, hi = np.load(tmp_path / f"range{rank}.npy")
        got[lo:hi] = np.load(tmp_path / f"shard{rank}.npy")
    assert np.array_equal(got, want)
