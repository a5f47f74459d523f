"""A device-side synchronisation timeout is REPORTED, not computed through.

blind_rotate_ll2_kernel (the default shape for launches of 257-1536 rotations and for tails) lets its four inverse
waves meet at a counter in LDS with a bounded wait.  When the wait expires the wave sets the device's fault word and
the host turns it into status -5 wherever it observes completion (Synchronize, StreamQuery, StreamSynchronize, the
scheduler's event polls); the C++ shim aborts on it like include/details/error_gpu.cuh:40-60.  The fault is injected by a
DIAGNOSTIC build (cufhe_amd/build.py: LL2_TIMEOUT -- one wave of workgroup 0 skips one arrival), loaded in a child
process through CUFHE_AMD_LIBRARY; the product build runs the same launch cleanly."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cufhe_amd as eng
from cufhe_amd import _lib
api = eng.api
assert (os.environ.get("CUFHE_AMD_LIBRARY") or "libcufhe_amd.so") in _lib.LIB_PATH
n, count = int(eng.PARAMS.n), 512                     # 512 rotations: one round of the paired kernel
rng = np.random.default_rng(9)
bk = rng.integers(0, 2**32, size=int(eng.PARAMS.bk_words), dtype=np.uint64).astype(np.uint32)
ksk = rng.integers(0, 2**32, size=int(eng.PARAMS.ksk_words), dtype=np.uint64).astype(np.uint32)
ins = [rng.integers(0, 2**32, size=(count, n + 1), dtype=np.uint64).astype(np.uint32) for _ in range(2)]
eng.SetGPUNum(1); eng.Initialize(bk, ksk)
d = [api.DeviceBuffer(x.size).upload(x) for x in ins]
out = api.DeviceBuffer(count * (n + 1))

def run(cnt):
    api.gate_batch(api.NAND, 0, out, d[0], d[1], count=cnt)
    eng.Synchronize()
    return out.download().reshape(count, n + 1)[:cnt].copy()

expect_fault = sys.argv[1] == "fault"
try:
    first = run(count)
    faulted = False
except _lib.CufheAmdError as e:
    faulted = True
    assert "error -5" in str(e) and "blind_rotate_ll2_kernel" in str(e), str(e)
assert faulted == expect_fault, "fault %s" % ("not reported" if expect_fault else "reported by a clean run")
if expect_fault:
    # sticky: every later completion reports it, also through StreamQuery on the per-gate API's scheduler
    try:
        eng.Synchronize(); raise SystemExit("fault status was not sticky")
    except _lib.CufheAmdError:
        pass
    st = eng.Stream(0); st.Create()
    q = _lib.lib.cufhe_amd_stream_query(0, st.st())
    assert q == -5, q
    # CleanUp + Initialize give a working device again; a launch that takes the single-rotation kernel is clean
    eng.CleanUp(); eng.Initialize(bk, ksk)
    small = run(8)
    eng.Synchronize()
    np.save(OUT, small)
else:
    np.save(OUT, first[:8])
eng.CleanUp()
print("child ok")
'''


def _build_mod():
    spec = importlib.util.spec_from_file_location("cufhe_amd_build", os.path.join(ROOT, "cufhe_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(tmp_path, mode, lib=None):
    env = dict(os.environ)
    env.pop("CUFHE_AMD_LIBRARY", None)
    if lib:
        env["CUFHE_AMD_LIBRARY"] = lib
    out = str(tmp_path / f"{mode}.npy")
    p = subprocess.run([sys.executable, "-c", f"ROOT={ROOT!r}\nOUT={out!r}\n" + CHILD, mode], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "child ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    return out


@pytest.mark.gpu
def test_ll2_rendezvous_timeout_is_reported(tmp_path):
    import numpy as np
    lib = _build_mod().build_fault_injection()
    clean = np.load(_run(tmp_path, "clean"))
    after = np.load(_run(tmp_path, "fault", lib))
    # after the fault was reported and the device re-initialised, the same gates give the product build's words
    assert np.array_equal(clean, after)


def test_fault_injection_is_fenced_off_the_product():
    """The injection switch compiles only into a diagnostic build, and the product library does not carry it."""
    src = open(os.path.join(ROOT, "cufhe_amd", "csrc", "fpfield.h")).read()
    assert "CUFHE_AMD_ABL_LL2_TIMEOUT" in src and "#error" in src
    mod = _build_mod()
    assert os.path.basename(mod.FAULT_OUT) != os.path.basename(mod.OUT)
