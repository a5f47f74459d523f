"""Words of full-size gates against a fixture that shares no code with the oracle or the kernels.

tests/golden/golden_independent_v2.json is written by tests/golden/make_golden_independent.py: pure numpy / Python
integers, the external product as an exact schoolbook negacyclic convolution (no transform), restated from the
reference's text.  Both the C oracle (CPU test) and the HIP path (GPU test) must reproduce its words: all ten two-input
gates, MUX and NMUX on level-0 ciphertexts and NAND on level-1 ciphertexts of the BASELINE set, one NAND through the
N = 2048 ring.  Keys are regenerated from the fixture's seeds and checked
against its sha256 sums -- a mismatch there is a failure, not a skip."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

FIXTURE = os.path.join(ol.ROOT, "tests", "golden", "golden_independent_v2.json")
OPS = {n: i for i, n in enumerate(ol.OPS)}


def words(count, rng, bits=32):
    if bits == 32:
        return rng.integers(0, 2**32, size=count, dtype=np.uint64).astype(np.uint32)
    return rng.integers(0, 2**64, size=count, dtype=np.uint64)


_keys = {}


def keys_for(case):
    """the generator's key words, from its seed; the same draw order as make_golden_independent.py"""
    ring, seed = case["ring"], case["key"]["seed"]
    if (ring, seed) not in _keys:
        rng = np.random.default_rng(seed)
        if ring == 1024:
            bk = words(630 * 6 * 2 * 1024, rng)
            ksk = words(1024 * 8 * 2 * 631, rng)
        else:
            bk = words(630 * 8 * 2 * 2048, rng, 64)
            ksk = words(2048 * 7 * 2 * 631, rng)
        assert hashlib.sha256(bk.tobytes()).hexdigest() == case["key"]["bk_sha256"], "numpy generated other key words than the fixture was made with"
        assert hashlib.sha256(ksk.tobytes()).hexdigest() == case["key"]["ksk_sha256"]
        _keys[(ring, seed)] = (bk, ksk)
    return _keys[(ring, seed)]


def fixture():
    return json.load(open(FIXTURE))


def cases():
    """(case, operand arrays, expected words)"""
    fx = fixture()
    for c in fx["cases"]:
        ins = [np.array(fx["inputs_level%d" % c["level"]][i], np.uint32) for i in c["operands"]]
        yield c, ins, np.array(c["expected"], np.uint32)


def test_fixture_is_what_the_generator_describes():
    fx = fixture()
    got = [(c["ring"], c["level"], c["op"]) for c in fx["cases"]]
    assert got == [(1024, 0, op) for op in ol.OPS[:12]] + [(1024, 1, "NAND"), (2048, 0, "NAND")]
    assert all(len(c["expected"]) == (1025 if c["level"] else 631) for c in fx["cases"])
    src = open(os.path.join(ol.ROOT, "tests", "golden", "make_golden_independent.py")).read()
    assert "import oracle" not in src and "cufhe_amd" not in src.split('"""')[2], "the generator must not share code with the oracle or the product"


def test_oracle_words_match_independent_generator(oracle):
    L = oracle
    for case, ins, want in cases():
        bk, ksk = keys_for(case)
        got = np.zeros(want.size, np.uint32)
        op = np.array([OPS[case["op"]]], np.int32)
        third = ins[2].ctypes.data if len(ins) > 2 else None
        if case["ring"] == 1024:
            ek = L.orc_evalkey_create(bk, ksk)
            L.orc_gate_batch(ek, op, 0, case["level"], 1, got, ins[0], ins[1].ctypes.data, third, 1)
            L.orc_evalkey_destroy(ek)
        else:
            ek = L.orc2_evalkey_create(bk, ksk)
            L.orc2_gate_batch(ek, op, 0, 1, got, ins[0], ins[1].ctypes.data, third, 1)
            L.orc2_evalkey_destroy(ek)
        assert np.array_equal(got, want), f"oracle words differ from the independent generator: ring {case['ring']} level {case['level']} {case['op']}"


GPU_CHILD = r'''
import json, os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_golden_independent as tg
import cufhe_amd as eng
api = eng.api
eng.SetGPUNum(1)
loaded = None
for case, ins, want in tg.cases():
    bk, ksk = tg.keys_for(case)
    if loaded != case["ring"]:
        if case["ring"] == 1024:
            eng.Initialize(bk, ksk)
        else:
            eng.Initialize()
            api.lvl2_initialize(bk, ksk)
        loaded = case["ring"]
    d = [api.DeviceBuffer(x.size).upload(x) for x in ins]
    out = api.DeviceBuffer(want.size)
    third = d[2] if len(d) > 2 else None
    if case["ring"] == 1024:
        api.gate_batch(tg.OPS[case["op"]], case["level"], out, d[0], d[1], third, count=1)
    else:
        api.lvl2_gate_batch(tg.OPS[case["op"]], out, d[0], d[1], third, count=1)
    eng.Synchronize()
    assert np.array_equal(out.download(), want), \
        f"HIP words differ from the independent generator: ring {case['ring']} level {case['level']} {case['op']}"
eng.CleanUp()
print("child ok")
'''


@pytest.mark.gpu
def test_gpu_words_match_independent_generator():
    """In a process of its own: the keys of the fixture replace whatever the session's engine holds."""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, "-c", f"ROOT={ol.ROOT!r}\n" + GPU_CHILD], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "child ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
