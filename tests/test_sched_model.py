"""The stream scheduler (cufhe_amd/csrc/sched_core.h) against a stubbed device layer, on the CPU.

tests/host/sched_harness.cpp drives the same C++ scheduler the HIP library uses through a fake
asynchronous device (in-order streams, events, late-executing copies, gates executed in random order
within a launch) and compares every observable result with a plain in-order interpreter of the
reference API (include/cufhe_gpu.cuh:193-313): random programs of copying gates, g-gates, explicit
copies, StreamQuery polls, host edits after Synchronize and ciphertexts destroyed mid-program, over
1-3 devices (the SetGPUNum(G > 1) routing of test/test_gate_gpu_multi.cc:36-93), plus the shaped
programs of tests/cpp/test_gate_api.cpp with their launch counts.
"""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "sched_harness.cpp")
EXE = os.path.join(ROOT, "tests", "host", "sched_harness")


@pytest.fixture(scope="module")
def harness():
    deps = [SRC, os.path.join(ROOT, "cufhe_amd", "csrc", "sched_core.h")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-o", EXE, SRC])
    return EXE


def _run(exe, seeds, gpus, threaded, rename=0, zero_copy=0, two_lane=1):
    out = subprocess.run([exe, str(seeds), str(gpus), str(threaded), str(rename), str(zero_copy), str(two_lane)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ALL PASS" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    return {s["name"]: s for s in (json.loads(l[6:]) for l in out.stdout.splitlines() if l.startswith("SHAPE "))}


@pytest.mark.parametrize("threaded", [0, 1])
def test_random_programs_match_in_order_semantics(harness, threaded):
    shapes = _run(harness, 120, 3, threaded)
    # test/test_api_gpu.cu:140-159: 64 chains x 5 in-place gates = 5 dependence levels, however the chains interleave
    assert shapes["chained"]["gates"] == 320 and shapes["chained"]["launch_sequences"] <= 6
    # 16 ripple-carry adders issued bit by bit: 4 levels per bit, batched across the adders
    assert shapes["ripple_adders"]["gates"] == 640 and shapes["ripple_adders"]["launch_sequences"] <= 40
    # test/test_intensive.cc: 800 gates on 200 polled streams run as a handful of launches, and the three
    # shared inputs are uploaded once, not once per gate
    assert shapes["intensive"]["launch_sequences"] <= 8 and shapes["intensive"]["uploads"] <= 8
    assert shapes["multi_gpu"]["gates"] == 192


def test_random_programs_with_output_renaming(harness):
    """DeviceSched::rename_outputs ("sched_rename"): outputs take fresh device buffers instead of waiting for the users of
    the old one.  Same in-order semantics on the random programs (1-3 devices, worker threads); the ripple-carry adders,
    whose temporaries are re-used bit after bit, drop from 4 levels per bit to 2."""
    shapes = _run(harness, 120, 3, 1, rename=1)
    assert shapes["ripple_adders"]["gates"] == 640 and shapes["ripple_adders"]["launch_sequences"] <= 18
    assert shapes["chained"]["launch_sequences"] <= 6 and shapes["intensive"]["launch_sequences"] <= 8


def test_random_programs_with_zero_copy_staging(harness):
    """Backend::device_alias ("sched_zero_copy", the HIP library's default): the scatter / gather kernels work on the pinned
    staging blocks themselves and the inputs of a flush's first level are scattered chunk by chunk while the launch worker is
    still gathering the rest.  Same in-order semantics on the random programs, with and without renaming."""
    shapes = _run(harness, 120, 3, 1, zero_copy=1)
    assert shapes["chained"]["launch_sequences"] <= 6 and shapes["intensive"]["uploads"] <= 8
    _run(harness, 60, 3, 1, rename=1, zero_copy=1)
    _run(harness, 60, 2, 0, zero_copy=1)


def test_random_netlists_scheduled_gate_by_gate_on_two_lanes(harness):
    """DeviceSched::compile_two_lane ("sched_two_lane"): a flush of several dependence levels is list-scheduled gate by gate onto a chain
    lane and a bulk lane (two internal streams) instead of level by level.  The harness's random netlists -- deep chains beside wide
    work, re-used temporaries, in-place gates, Mux, fetched results, a second flush depending on the first -- are made for it: hundreds of
    flushes take that path under the stub's cost model, every value equals the in-order interpreter's, and the ripple-carry adders do too."""
    out = subprocess.run([harness, "150", "3", "1", "1", "0", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ALL PASS" in out.stdout, out.stdout[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("random netlists:")][0]
    assert int(line.split("failures, ")[1].split()[0]) >= 100, line            # the path was taken
    shapes = {s["name"]: s for s in (json.loads(l[6:]) for l in out.stdout.splitlines() if l.startswith("SHAPE "))}
    assert shapes["ripple_adders"]["two_lane_groups"] == 1 and shapes["ripple_adders"]["failures"] == 0
    _run(harness, 100, 1, 0, rename=1, zero_copy=1, two_lane=1)
    _run(harness, 100, 2, 1, rename=1, two_lane=0)        # the same netlists level by level


def test_random_programs_on_one_device_without_workers(harness):
    """One device, launches on the issuing thread: the configuration in which a caller's own work on a raw stream handle (Stream::st():
    reads and writes of a ciphertext's device buffer between recorded gates) meets the densest program."""
    _run(harness, 200, 1, 0)
    _run(harness, 200, 2, 0, rename=1, zero_copy=1)


def test_sanitizers(harness, tmp_path):
    """The same run under AddressSanitizer + UBSan (CPU build only), worker threads on."""
    exe = str(tmp_path / "sched_harness_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-o", exe, SRC])
    _run(exe, 25, 3, 1)
    _run(exe, 25, 3, 1, rename=1)
    _run(exe, 25, 3, 1, zero_copy=1)


def test_thread_sanitizer(harness, tmp_path):
    """The issuing thread against the per-device launch workers under ThreadSanitizer (CPU build only)."""
    exe = str(tmp_path / "sched_harness_tsan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-o", exe, SRC])
    for rename, zero_copy in (("0", "0"), ("1", "0"), ("0", "1")):
        out = subprocess.run([exe, "12", "3", "1", rename, zero_copy], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "ALL PASS" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
        assert "ThreadSanitizer" not in out.stderr, out.stderr[:4000]


# (what is broken, the exact source text, its replacement, harness arguments under which the break must show)
MUTATIONS = [
    ("the caller's stream no longer waits for the gates issued before Stream::st()",
     "if (int rc = be_->caller_stream_wait(stream, g->done->ev)) return fail(rc, be_->error_text());", ";", ["120", "3", "1", "1", "0"]),
    ("gates no longer wait for what the caller put on the raw handle before them",
     "if (rc == 0) step(be_->wait_for_caller_stream(s, cs));", ";", ["200", "1", "0", "0", "0"]),
    ("a device buffer the caller may have rewritten through the raw handle still stands for the result on its way to tlwehost",
     "fence_epoch(c->host_stream) == c->host_epoch", "true", ["200", "1", "0", "0", "0"]),
    ("a flush no longer waits for the flush on another internal stream that produces its inputs",
     "g->deps.push_back(dg->done);", ";", ["120", "3", "1", "0", "0"]),
    ("a write no longer follows the recorded readers of the buffer it overwrites",
     "if (has_readers(po)) D = std::max(D, max_reader(po) + 1);         // write after read", ";", ["120", "3", "1", "0", "0"]),
    ("a launch of one lane no longer waits for the other lane's launch that produces its operands",
     "if (ev) step(be_->stream_wait(st, ev));", ";", ["150", "1", "0", "1", "0", "1"]),
    ("the remainder of a two-lane flush no longer waits for the chain lane",
     "if (void* joined = new_event(sc)) step(be_->stream_wait(s, joined));", ";", ["150", "1", "0", "1", "0", "1"]),
    ("with per-gate scheduling on, an output whose buffer still has recorded readers is no longer renamed",
     "for_readers(po, [&](uint32_t r) { users = users || r >= base_depth_; });", ";", ["150", "1", "0", "1", "0", "1"]),
    ("with per-gate scheduling on, an output whose buffer holds an unrelated recorded write is no longer renamed",
     "if (!is_input && po.wdepth >= base_depth_) users = true;", ";", ["150", "1", "0", "1", "0", "1"]),
    ("a copy home runs among the other gates of a two-lane flush instead of behind them",
     "if (d.home_copy) { is_post[id] = 1; continue; }", "", ["150", "3", "0", "1", "0", "1"]),
    ("a renamed value refreshed by an upload on another stream is not copied home for that stream's StreamQuery",
     "&& !(pd.ustream == only_stream && only_stream != nullptr)", "", ["150", "3", "1", "1", "0"]),
]


@pytest.mark.parametrize("what,old,new,args", MUTATIONS, ids=[m[0][:40] for m in MUTATIONS])
def test_harness_notices_a_broken_scheduler(tmp_path, what, old, new, args):
    """The harness is only worth something if a scheduler with one ordering edge removed FAILS it: each mutation deletes one edge of
    sched_core.h (a copy; the tree is not touched) and the random programs must report mismatches."""
    core = open(os.path.join(ROOT, "cufhe_amd", "csrc", "sched_core.h")).read()
    assert core.count(old) == 1, "mutation target moved: " + old
    (tmp_path / "cufhe_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "tests" / "host").mkdir(parents=True)
    (tmp_path / "cufhe_amd" / "csrc" / "sched_core.h").write_text(core.replace(old, new))
    (tmp_path / "tests" / "host" / "sched_harness.cpp").write_text(open(SRC).read())
    exe = str(tmp_path / "mutant")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-o", exe, str(tmp_path / "tests" / "host" / "sched_harness.cpp")])
    out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900)
    assert out.returncode != 0 and "FAIL" in out.stdout, "the harness passed a scheduler in which " + what
