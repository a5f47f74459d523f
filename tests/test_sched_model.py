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


def _run(exe, seeds, gpus, threaded, rename=0, zero_copy=0):
    out = subprocess.run([exe, str(seeds), str(gpus), str(threaded), str(rename), str(zero_copy)], capture_output=True, text=True, timeout=900)
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
