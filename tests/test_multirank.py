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
# the second timed workload of a plain multi-rank bench run (extra_workloads.mixed_32768_strong): a strong split of ANOTHER gate list
# in the same processes, every rank's share gathered on all ranks
total2 = 13
lo2, hi2 = d.shard(total2, rank, world)
ins2 = [keys.encrypt(np.random.default_rng(6 + i).integers(0, 2, size=total2).astype(np.uint8), 0, seed=400 + i) for i in range(2)]
ops2 = np.array([[3, 4, 5, 0][g % 4] for g in range(total2)], np.int32)
dist.barrier(); t0 = time.perf_counter()
out2 = keys.gate_batch(ops2[lo2:hi2], 0, ins2[0][lo2:hi2], ins2[1][lo2:hi2])
own = time.perf_counter() - t0
dist.barrier(); el2 = d.max_over_ranks(time.perf_counter() - t0, dist)
per = d.gather_objects({"rank": rank, "first_gate": lo2, "gates_per_step": hi2 - lo2, "ms_per_step": 1e3 * own}, dist)
assert [p["rank"] for p in per] == list(range(world)) and sum(p["gates_per_step"] for p in per) == total2
assert all(per[i]["first_gate"] + per[i]["gates_per_step"] == per[i + 1]["first_gate"] for i in range(world - 1))
assert el2 >= max(p["ms_per_step"] for p in per) * 1e-3 * 0.999
np.save(os.path.join(OUT, f"second{rank}.npy"), out2)
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
    # the strong split of the second workload: union of the ranks' shares == the single-rank words
    total2 = 13
    ins2 = [keys.encrypt(np.random.default_rng(6 + i).integers(0, 2, size=total2).astype(np.uint8), 0, seed=400 + i) for i in range(2)]
    ops2 = np.array([[3, 4, 5, 0][g % 4] for g in range(total2)], np.int32)
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"second{r}.npy") for r in range(2)]), keys.gate_batch(ops2, 0, ins2[0], ins2[1]))
    # bench.py wires it in: a plain multi-rank NAND run times configs[2] in the same processes, on every rank (barriers inside)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "if WORLD > 1 and not strong and wl == \"nand\" and not args.no_extra:\n        mixed_strong = mixed_strong_extra()" in src
    assert src.index("mixed_strong = mixed_strong_extra()") < src.index("if RANK == 0:\n        gates_per_step")


def _dist():
    import importlib.util
    spec = importlib.util.spec_from_file_location("d", os.path.join(ROOT, "cufhe_amd", "dist.py"))
    d = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(d)
    return d


STUB_CHILD = r'''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
mode = sys.argv[1]
if mode == "fail" and rank == 1:
    sys.exit(7)
if mode == "fail" or (mode == "hang" and rank == 0):
    time.sleep(600)                       # a rank waiting in a barrier for the one that died / a hung rank
print("noise on stdout of rank %d" % rank)
if rank == 0:
    print(json.dumps({"n_gpus": world, "port": os.environ["MASTER_PORT"]}))
'''


def test_self_launcher_with_stub_children():
    """bench.py --gpus N run plainly: spawn_ranks starts N fresh rank processes, relays rank 0's
    stdout, and a failing (or hung) rank ends the job with a non-zero exit code, not a hang."""
    import json
    import time
    d = _dist()
    rc, out = d.spawn_ranks([sys.executable, "-c", STUB_CHILD, "ok"], 3)
    assert rc == 0
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 3        # rank 0's line only
    assert out.count("noise") == 1
    t0 = time.monotonic()
    rc, out = d.spawn_ranks([sys.executable, "-c", STUB_CHILD, "fail"], 3)
    assert rc == 7 and time.monotonic() - t0 < 60                          # rank 1's code; ranks 0 and 2 were ended
    t0 = time.monotonic()
    rc, out = d.spawn_ranks([sys.executable, "-c", STUB_CHILD, "hang"], 2, timeout=2.0)
    assert rc == 124 and time.monotonic() - t0 < 60


LAUNCHER_UNDER_SIGTERM = r'''
import importlib.util, os, sys
spec = importlib.util.spec_from_file_location("d", os.path.join(ROOT, "cufhe_amd", "dist.py"))
d = importlib.util.module_from_spec(spec); spec.loader.exec_module(d)
child = "import os, time; open(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid())); time.sleep(600)" % OUT
open(os.path.join(OUT, "launcher"), "w").write(str(os.getpid()))
d.spawn_ranks([sys.executable, "-c", child], 2, timeout=500)
'''


def test_launcher_ends_its_ranks_when_it_is_terminated(tmp_path):
    """The ranks lead their own sessions, so a SIGTERM aimed at the launcher (a harness timeout) does not reach them by itself:
    spawn_ranks turns it into an exception and ends the ranks before it goes on -- no GPU-resident process is left behind."""
    import signal
    import time
    code = f"ROOT={ROOT!r}\nOUT={str(tmp_path)!r}\n" + LAUNCHER_UNDER_SIGTERM
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    deadline = time.monotonic() + 60
    while time.monotonic() < deadline and not all((tmp_path / f).exists() and (tmp_path / f).read_text() for f in ("pid0", "pid1")):
        time.sleep(0.1)
    pids = [int((tmp_path / f).read_text()) for f in ("pid0", "pid1")]
    p.send_signal(signal.SIGTERM)
    p.wait(60)
    assert p.returncode != 0
    for _ in range(100):
        alive = []
        for pid in pids:
            try:
                os.kill(pid, 0)
                alive.append(pid)
            except ProcessLookupError:
                pass
        if not alive:
            break
        time.sleep(0.1)
    assert not alive, f"ranks {alive} survived their launcher"


SHARED_GPU_CHILD = r'''
import json, os, sys, importlib.util
spec = importlib.util.spec_from_file_location("d", os.path.join(ROOT, "cufhe_amd", "dist.py"))
d = importlib.util.module_from_spec(spec); spec.loader.exec_module(d)
rank, local_rank, world = d.rank_env()
import torch.distributed as dist
dist.init_process_group("gloo", rank=rank, world_size=world)
mode, allow = sys.argv[1], sys.argv[2] == "allow"
# what cufhe_amd.api.device_identity(0) reports: ranks wrapped onto one GPU report the same PCI function and UUID
same = {"pci": "0000:05:00.0", "uuid": "aa" * 16, "hip_device": "0"}
mine = same if mode == "shared" else {"pci": "0000:%02x:00.0" % (5 + rank), "uuid": "%02x" % rank * 16, "hip_device": str(rank)}
ids = d.gather_objects(mine, dist)
try:
    distinct, shared = d.check_distinct_gpus(ids, allow)
except d.SharedGpuError as e:
    if rank == 0:
        sys.stderr.write("refused: %s\\n" % e)
    dist.barrier(); dist.destroy_process_group()
    sys.exit(3)
reports = d.gather_objects({"rank": rank, "gpu": mine, "value": 100.0 + rank, "ms_per_step": 40.0 - rank}, dist)
if rank == 0:
    line = {"n_gpus": distinct, "ranks": world}
    line.update(d.rank_summary(reports, allow))
    print(json.dumps(line))
dist.barrier(); dist.destroy_process_group()
'''


def test_ranks_sharing_a_gpu_are_refused_unless_allowed():
    """The multi-GPU bench line proves itself: every rank reports its physical GPU, rank 0 prints per_rank / distinct_gpus,
    and ranks that share a GPU end the run non-zero unless --allow-shared-gpu (then shared_gpu: true, n_gpus = distinct)."""
    import json
    d = _dist()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    code = f"ROOT={ROOT!r}\n" + SHARED_GPU_CHILD
    rc, out = d.spawn_ranks([sys.executable, "-c", code, "shared", "refuse"], 2, env=env, timeout=300)
    assert rc == 3 and not [l for l in out.splitlines() if l.startswith("{")]
    rc, out = d.spawn_ranks([sys.executable, "-c", code, "shared", "allow"], 2, env=env, timeout=300)
    assert rc == 0
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert line["shared_gpu"] is True and line["n_gpus"] == 1 and line["distinct_gpus"] == 1 and line["ranks"] == 2
    assert [r["rank"] for r in line["per_rank"]] == [0, 1] and line["value_min_rank"] == 100.0 and line["value_max_rank"] == 101.0
    rc, out = d.spawn_ranks([sys.executable, "-c", code, "distinct", "refuse"], 2, env=env, timeout=300)
    assert rc == 0
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert "shared_gpu" not in line and line["n_gpus"] == 2 and line["distinct_gpus"] == 2
    # the pure function, and the identity fallback when the runtime reports no UUID
    assert d.check_distinct_gpus([{"pci": "a", "uuid": "?"}, {"pci": "b", "uuid": "?"}]) == (2, False)
    assert d.check_distinct_gpus(["x", "x", "y"], allow_shared=True) == (2, True)
    import pytest as _pt
    with _pt.raises(d.SharedGpuError):
        d.check_distinct_gpus(["x", "x"])
    # bench.py wires it in: gathered before any work, exit code 3, flag documented
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("check_distinct_gpus(identities") < src.index("eng.Initialize(bk, ksk)")
    assert "--allow-shared-gpu" in src and "rank_summary(reports" in src


def test_bench_refuses_world_mismatch_and_self_launches_before_torch():
    """The launcher branch of bench.py runs before torch / the HIP library are imported (a process
    that touched the GPU must not start ranks), and a WORLD_SIZE that contradicts --gpus is an error."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("self_launch(ARGS)") < src.index("import torch")
    assert src.index("self_launch(ARGS)") < src.index("import numpy")
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


GPU_WORKER = r'''
import os, sys, importlib.util
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("d", os.path.join(ROOT, "cufhe_amd", "dist.py"))
d = importlib.util.module_from_spec(spec); spec.loader.exec_module(d)
rank, local_rank, world = d.rank_env()
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = d.device_for_rank(local_rank, torch.cuda.device_count())      # wraps on a 1-GPU box
torch.cuda.set_device(dev)
import oracle_lib as ol
import cufhe_amd as eng
eng.api.set_option("device_base", dev)
L = ol.load(); keys = ol.Keys(L, seed=1)
eng.SetGPUNum(1); eng.Initialize(keys.bk, keys.ksk)
count = 24
bits = np.random.default_rng(5).integers(0, 2, size=(2, count)).astype(np.uint8)
ins = [keys.encrypt(bits[i], 0, seed=300 + i) for i in range(2)]
ops = np.array([[3, 4, 5, 0][g % 4] for g in range(count)], np.int32)
lo, hi = d.shard(count, rank, world)
n = hi - lo
d0 = eng.api.DeviceBuffer(n * (ol.n + 1)).upload(np.ascontiguousarray(ins[0][lo:hi]))
d1 = eng.api.DeviceBuffer(n * (ol.n + 1)).upload(np.ascontiguousarray(ins[1][lo:hi]))
dout = eng.api.DeviceBuffer(n * (ol.n + 1))
dist.barrier()
eng.api.gate_batch(np.ascontiguousarray(ops[lo:hi]), 0, dout, d0, d1, None, count=n, device=0)
eng.Synchronize()
dist.barrier()
np.save(os.path.join(OUT, f"shard{rank}.npy"), dout.download().reshape(n, -1))
np.save(os.path.join(OUT, f"range{rank}.npy"), np.array([lo, hi]))
mapped = [l.split()[-1] for l in open("/proc/self/maps") if "libcufhe_amd.so" in l]
assert mapped, "HIP library not mapped in the rank"
eng.CleanUp()
dist.barrier(); dist.destroy_process_group()
'''


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_hip_library_words(tmp_path, keys):
    """Two ranks, each loading libcufhe_amd.so with its own `device_base` (wrapping onto the one GPU of
    a 1-GPU box), each running its contiguous shard of a mixed gate list through the C ABI: the
    union of the shards == the oracle's words (test/test_gate_gpu_multi.cc:36-93 shards by stream
    device; include/cufhe_gpu.cuh:154-159; per-GPU key replicas src/bootstrap_gpu.cu:115-137)."""
    d = _dist()
    code = f"ROOT={ROOT!r}\nOUT={str(tmp_path)!r}\n" + GPU_WORKER
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HIP_VISIBLE_DEVICES"):
        env.pop(k, None)
    rc, out = d.spawn_ranks([sys.executable, "-c", code], 2, env=env, timeout=900)
    assert rc == 0, out
    count = 24
    bits = np.random.default_rng(5).integers(0, 2, size=(2, count)).astype(np.uint8)
    ins = [keys.encrypt(bits[i], 0, seed=300 + i) for i in range(2)]
    ops = np.array([[3, 4, 5, 0][g % 4] for g in range(count)], np.int32)
    want = keys.gate_batch(ops, 0, ins[0], ins[1])
    got = np.zeros_like(want)
    for rank in range(2):
        lo, hi = np.load(tmp_path / f"range{rank}.npy")
        got[lo:hi] = np.load(tmp_path / f"shard{rank}.npy")
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_bench_self_launch_two_ranks_on_this_box():
    """`python bench.py --gpus 2` run plainly on a box with ONE GPU: refused (both ranks wrap onto the same device: the line
    would claim two GPUs); with --allow-shared-gpu it runs as a labelled rehearsal: shared_gpu, n_gpus = 1, both ranks listed
    with the same physical GPU and their own rates."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    import cufhe_amd as eng
    one_gpu = eng.api.DeviceCount() == 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--gates", "512"]
    if one_gpu:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 3 and "distinct GPU" in p.stderr, (p.returncode, p.stderr[-2000:])
        assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    p = subprocess.run(cmd + ["--allow-shared-gpu"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["ranks"] == 2 and lines[0]["value"] > 0
    line = lines[0]
    assert len(line["per_rank"]) == 2 and all(r["value"] > 0 and r["gpu"]["pci"] != "?" for r in line["per_rank"])
    assert line["distinct_gpus"] == line["n_gpus"] == len({r["gpu"]["uuid"] for r in line["per_rank"]})
    if one_gpu:
        assert line["shared_gpu"] is True and line["n_gpus"] == 1
    assert line["value_min_rank"] <= line["value_max_rank"]
    # BASELINE configs[2] in the same run: 32 768 mixed gates split contiguously over the two ranks, words checked on rank 0
    ms = line["extra_workloads"]["mixed_32768_strong"]
    assert ms["value"] > 0 and ms["scaling"] == "strong" and ms["gpu_words_match_oracle"] is True
    assert [r["gates_per_step"] for r in ms["per_rank"]] == [16384, 16384] and [r["first_gate"] for r in ms["per_rank"]] == [0, 16384]
    assert abs(line["summary"]["mixed_32768_strong_gates_per_s"] - ms["value"]) < 1e-3        # (the summary rounds to four decimals)
    assert list(line)[-1] == "summary" and line["summary"]["word_checks"]["failed"] == 0          # last key: the tail of the line shows it
    dumped = json.dumps(line)
    assert len(dumped) - dumped.index('"summary"') < 2048        # ... all of it within the last 2 KB


@pytest.mark.gpu
def test_bench_api_mode_one_process_many_devices():
    """`bench.py --mode api --gpus G`: one process, SetGPUNum(G), Streams round-robin the devices (the reference's multi-GPU
    shape, test/test_gate_gpu_multi.cc:36-93).  On fewer GPUs than G it is refused unless --allow-shared-gpu."""
    import json
    import cufhe_amd as eng
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "api", "--gpus", "2", "--steps", "2", "--warmup", "1", "--gates", "1024"]
    if eng.api.DeviceCount() < 2:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    p = subprocess.run(cmd + ["--allow-shared-gpu"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert line["config"]["mode"] == "api" and line["config"]["logical_devices"] == 2 and line["value"] > 0
    assert len(line["per_device"]) == 2 and sum(d["gates"] for d in line["per_device"]) == 2048
    assert all(d["gates"] == 1024 for d in line["per_device"])          # streams round-robin the devices
    assert line["n_gpus"] == line["distinct_gpus"] and (line["distinct_gpus"] == 2 or line["shared_gpu"] is True)
