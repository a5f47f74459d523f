"""Words of full-size gates against a fixture that shares no code with the oracle or the kernels.

tests/golden/golden_independent_v5.json is written by tests/golden/make_golden_independent.py: pure numpy / Python
integers, the external product as an exact schoolbook negacyclic convolution (no transform), restated from the
reference's text.  Both the C oracle (CPU test) and the HIP path (GPU test) must reproduce its words: all ten two-input
gates, MUX and NMUX on level-0 ciphertexts and NAND on level-1 ciphertexts of the BASELINE set, one NAND through the
N = 2048 ring, and a few gates on each of the other compiled parameter sets (k = 2 / N = 512; n = 500 / l = 2 / Bg = 2^10); since v4 also
the corner inputs the oracle tests use (runs of abar = 0, bbar = 2N / N / 1, words 0x7FFFFFFF, a key of extreme words), MUX on level-1
ciphertexts and Not / Copy; since v5 the TRLWE-level primitives (CMUXNTT, bootstrap to a TRLWE, sample extract + key switch, Refresh).
tests/golden/golden_independent_smallmod_v1.json (the same script, argument `smallmod`) holds the small-modulus mode: the BASELINE
numbers as the reference computes them when built with -DUSE_SMALL_NTT_MODULUS (include/ntt_gpu/ntt_small_modulus.cuh), the product
mod P as a schoolbook sum in exact integers reduced with Python's %.
Keys are regenerated from the fixture's seeds and checked
against its sha256 sums -- a mismatch there is a failure, not a skip."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

FIXTURE = os.path.join(ol.ROOT, "tests", "golden", "golden_independent_v5.json")
FIXTURE_SMALLMOD = os.path.join(ol.ROOT, "tests", "golden", "golden_independent_smallmod_v1.json")
# round 6 (the same script, argument `cmux_sets`): CMUXNTT on the other compiled sets -- uniform and extreme TRGSW words, chained in place
FIXTURE_CMUX_SETS = os.path.join(ol.ROOT, "tests", "golden", "golden_independent_cmux_sets_v1.json")
OPS = {n: i for i, n in enumerate(ol.OPS)}


def words(count, rng, bits=32):
    if bits == 32:
        return rng.integers(0, 2**32, size=count, dtype=np.uint64).astype(np.uint32)
    return rng.integers(0, 2**64, size=count, dtype=np.uint64)


_keys = {}


# (n, N, k, l, bits, t) of the sets of the fixture: the draw sizes of the generator's key_for()
SETS = {"default": (630, 1024, 1, 3, 32, 8), "lvl2": (630, 2048, 1, 4, 64, 7), "k2n512": (630, 512, 2, 3, 32, 8), "cggi16": (500, 1024, 1, 2, 32, 8),
        "smallmod": (630, 1024, 1, 3, 32, 8)}


def keys_for(case):
    """the generator's key words, from its seed; the same draw order as make_golden_independent.py"""
    name, seed = case["set"], case["key"]["seed"]
    if (name, seed) not in _keys:
        n, N, k, l, bits, t = SETS[name]
        rng = np.random.default_rng(seed)
        if case["key"].get("kind") == "extreme":     # the generator's key_extreme(): first two CMux steps -2^31, the rest from five extreme words
            ext = np.array((0x80000000, 0x7FFFFFFF, 0, 0xFFFFFFFF, 0x80000001), np.uint32)
            bk = ext[rng.integers(0, ext.size, n * (k + 1) * l * (k + 1) * N)]
            bk[: 2 * (k + 1) * l * (k + 1) * N] = 0x80000000
        else:
            bk = words(n * (k + 1) * l * (k + 1) * N, rng, bits)
        ksk = words(k * N * t * 2 * (n + 1), rng)
        assert hashlib.sha256(bk.tobytes()).hexdigest() == case["key"]["bk_sha256"], "numpy generated other key words than the fixture was made with"
        assert hashlib.sha256(ksk.tobytes()).hexdigest() == case["key"]["ksk_sha256"]
        _keys[(name, seed)] = (bk, ksk)
    return _keys[(name, seed)]


def fixture():
    return json.load(open(FIXTURE))


def fixture_smallmod():
    return json.load(open(FIXTURE_SMALLMOD))


def cases():
    """(case, operand arrays, expected words), both fixtures"""
    for fx in (fixture(), fixture_smallmod()):
        for c in fx["cases"]:
            ins = [np.array(fx["inputs"][c["inputs"]][i], np.uint32) for i in c["operands"]]
            yield c, ins, np.array(c["expected"], np.uint32)


def cmux_set_cases():
    """(case, trgsw torus words, c1, c0, expected) of the CMUXNTT fixture on the other parameter sets; the TRGSWs are regenerated from
    the fixture's seed (the generator's cmux_trgsws()) and checked against its hashes"""
    fx = json.load(open(FIXTURE_CMUX_SETS))
    ext = np.array((0x80000000, 0x7FFFFFFF, 0, 0xFFFFFFFF, 0x80000001), np.uint32)
    for c in fx["cases"]:
        n, N, k, l, bits, t = SETS[c["set"]]
        meta = fx["inputs"]["trgsw_" + c["set"]]
        rng = np.random.default_rng(meta["seed"])
        size = (k + 1) * l * (k + 1) * N
        uniform = words(size, rng)
        extreme = ext[rng.integers(0, ext.size, size)]
        extreme[: (k + 1) * N] = 0x80000000
        tg = (uniform, extreme)[c["trgsw"]]
        assert hashlib.sha256(tg.tobytes()).hexdigest() == meta["sha256"][c["trgsw"]], "numpy generated other TRGSW words than the fixture was made with"
        trl = [np.array(x, np.uint32) for x in fx["inputs"][c["inputs"]]]
        yield c, (uniform, extreme), [trl[i] for i in c["operands"]], np.array(c["expected"], np.uint32)


def test_oracle_cmux_on_the_other_sets_matches_independent_generator():
    """orc_cmux of the per-set oracle builds (what the GPU's CMUXNTT on a set is compared with everywhere else) == the schoolbook sums."""
    seen = []
    for c, (uniform, extreme), (c1, c0), want in cmux_set_cases():
        L = ol.load_set(c["set"])
        got = np.zeros(want.size, np.uint32)
        if c["op"] == "CMUXNTT_CHAINED":      # first = cmux(uniform, c0, c1); then cmux(extreme, c1 = operand 0, c0 = first), in place on `first`
            first = np.zeros(want.size, np.uint32)
            L.orc_cmux(first, uniform, c0, c1)
            L.orc_cmux(first, extreme, c1, first.copy())
            got = first
        else:
            L.orc_cmux(got, (uniform, extreme)[c["trgsw"]], c1, c0)
        assert np.array_equal(got, want), f"oracle CMUXNTT differs from the independent generator: {c['set']} {c['op']} trgsw {c['trgsw']}"
        seen.append((c["set"], c["op"], c["trgsw"]))
    assert seen == [(s, op, tg) for s in ("k2n512", "cggi16") for op, tg in (("CMUXNTT", 0), ("CMUXNTT", 1), ("CMUXNTT_CHAINED", 1))]


def test_fixture_is_what_the_generator_describes():
    fx = fixture()
    got = [(c["set"], c["level"], c["op"]) for c in fx["cases"]]
    assert got == [("default", 0, op) for op in ol.OPS[:12]] + [("default", 1, "NAND"),
                                                                ("default", 0, "NAND"), ("default", 0, "AND"), ("default", 0, "OR"),      # corner inputs
                                                                ("default", 1, "MUX"), ("default", 0, "NOT"), ("default", 1, "COPY"),
                                                                ("default", 2, "CMUXNTT"), ("default", 2, "BOOT2TRLWE"), ("default", 2, "SEIKS"), ("default", 2, "REFRESH"),
                                                                ("default", 0, "NAND"),                                                   # extreme key words
                                                                ("k2n512", 0, "NAND"), ("k2n512", 0, "XOR"),
                                                                ("k2n512", 0, "MUX"), ("k2n512", 1, "NAND"), ("cggi16", 0, "NAND"), ("cggi16", 0, "ORYN"),
                                                                ("cggi16", 1, "XOR"), ("lvl2", 0, "NAND")]
    assert [c["inputs"] for c in fx["cases"][13:16]] == ["level0_edge_a", "level0_edge_b", "level0_edge_c"] and fx["cases"][23]["key"]["kind"] == "extreme"
    # the corner inputs sit where the generator says (the reference's modswitch, include/gatebootstrapping_gpu.cuh:10-16, on python integers)
    e = {t: [np.array(x, np.uint32).astype(np.int64) for x in fx["inputs"]["level0_edge_" + t]] for t in "abc"}
    ms = lambda v: (int(v) & 0xFFFFFFFF) >> 21
    ca = (-e["a"][0] - e["a"][1]) & 0xFFFFFFFF
    assert all(ms(ca[i] + (1 << 20)) == 0 for i in list(range(8)) + [100, 629]) and 2048 - ms(ca[630] + (1 << 29)) == 2048
    cb = (e["b"][0] + e["b"][1]) & 0xFFFFFFFF
    assert all(ms(cb[i] + (1 << 20)) == 1024 for i in range(8)) and 2048 - ms(cb[630] - (1 << 29)) == 1024
    cc = (e["c"][0] + e["c"][1]) & 0xFFFFFFFF
    assert all(ms(cc[i] + (1 << 20)) == 0 for i in range(4)) and 2048 - ms(cc[630] + (1 << 29)) == 1
    for c in fx["cases"]:
        n, N, k = SETS[c["set"]][:3]
        if c["level"] == 2:          # TRLWE-level primitives: a TRLWE out, except sample extract + key switch
            assert len(c["expected"]) == (n + 1 if c["op"] == "SEIKS" else (k + 1) * N)
            continue
        assert len(c["expected"]) == (k * N + 1 if c["level"] else n + 1)
    fs = fixture_smallmod()
    assert [(c["set"], c["level"], c["op"], c["inputs"]) for c in fs["cases"]] == [("smallmod", 0, "NAND", "level0"), ("smallmod", 0, "XOR", "level0"),
                                                                                  ("smallmod", 0, "MUX", "level0"), ("smallmod", 1, "NAND", "level1"),
                                                                                  ("smallmod", 0, "NAND", "level0_edge_a")]
    assert all(len(c["expected"]) == (1025 if c["level"] else 631) for c in fs["cases"])
    src = open(os.path.join(ol.ROOT, "tests", "golden", "make_golden_independent.py")).read()
    assert "import oracle" not in src and "cufhe_amd" not in src.split('"""')[2], "the generator must not share code with the oracle or the product"


def test_oracle_words_match_independent_generator(oracle):
    for case, ins, want in cases():
        bk, ksk = keys_for(case)
        got = np.zeros(want.size, np.uint32)
        if case["level"] == 2:       # TRLWE-level primitives (oracle/tfhe_oracle.h)
            ek = oracle.orc_evalkey_create(bk, ksk)
            if case["op"] == "CMUXNTT":
                oracle.orc_cmux(got, np.array(fixture()["inputs"]["trgsw"][0], np.uint32), ins[0], ins[1])
            elif case["op"] == "BOOT2TRLWE":
                oracle.orc_blind_rotate(ek, got, ins[0], -1)
            elif case["op"] == "SEIKS":
                oracle.orc_sample_extract_keyswitch(ek, got, ins[0])
            else:
                oracle.orc_refresh(ek, got, ins[0])
            oracle.orc_evalkey_destroy(ek)
            assert np.array_equal(got, want), f"oracle words differ from the independent generator: {case['op']}"
            continue
        op = np.array([OPS[case["op"]]], np.int32)
        third = ins[2].ctypes.data if len(ins) > 2 else None
        second = ins[1].ctypes.data if len(ins) > 1 else None
        if case["set"] == "lvl2":
            ek = oracle.orc2_evalkey_create(bk, ksk)
            oracle.orc2_gate_batch(ek, op, 0, 1, got, ins[0], second, third, 1)
            oracle.orc2_evalkey_destroy(ek)
        else:
            L = ol.load_set(case["set"])           # the oracle compiled for the set ("default" = liboracle.so)
            ek = L.orc_evalkey_create(bk, ksk)
            L.orc_gate_batch(ek, op, 0, case["level"], 1, got, ins[0], second, third, 1)
            L.orc_evalkey_destroy(ek)
        assert np.array_equal(got, want), f"oracle words differ from the independent generator: {case['set']} level {case['level']} {case['op']}"


GPU_CHILD = r'''
import json, os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_golden_independent as tg
import cufhe_amd as eng
api = eng.api
eng.SetGPUNum(1)
ps_index = {api.ps_params(i).name.decode(): i for i in range(api.ps_count())}
loaded = None
for case, ins, want in tg.cases():
    bk, ksk = tg.keys_for(case)
    name = case["set"]
    if loaded != (name, case["key"]["seed"]):
        if name == "default":
            eng.Initialize(bk, ksk)
        elif name == "lvl2":
            eng.Initialize()
            api.lvl2_initialize(bk, ksk)
        else:
            eng.Initialize()
            api.ps_initialize(ps_index[name], bk, ksk)
        loaded = (name, case["key"]["seed"])
    d = [api.DeviceBuffer(x.size).upload(x) for x in ins]
    out = api.DeviceBuffer(want.size)
    if case["level"] == 2:           # TRLWE-level primitives through the device-pointer batch entries
        if case["op"] == "CMUXNTT":
            trgsw_words = np.array(tg.fixture()["inputs"]["trgsw"][0], np.uint32)
            dtg = api.DeviceBuffer(trgsw_words.size).upload(trgsw_words)
            dntt = api.DeviceBuffer(2 * trgsw_words.size)        # NTT-domain doubles: two words each
            api.trgsw_to_ntt_batch(dtg, dntt, 1)
            api.cmux_batch(dntt, d[0], d[1], out, 1)
        elif case["op"] == "BOOT2TRLWE":
            api.blind_rotate_batch(d[0], out, 1)
        elif case["op"] == "SEIKS":
            api.sample_extract_keyswitch_batch(d[0], out, 1)
        else:
            api.refresh_batch(d[0], out, 1)
        eng.Synchronize()
        assert np.array_equal(out.download(), want), f"HIP words differ from the independent generator: {case['op']}"
        continue
    second = d[1] if len(d) > 1 else None
    third = d[2] if len(d) > 2 else None
    op = tg.OPS[case["op"]]
    if name == "default":
        api.gate_batch(op, case["level"], out, d[0], second, third, count=1)
    elif name == "lvl2":
        api.lvl2_gate_batch(op, out, d[0], second, third, count=1)
    else:
        api.ps_gate_batch(ps_index[name], op, out, d[0], second, third, count=1, level=case["level"])
    eng.Synchronize()
    assert np.array_equal(out.download(), want), \
        f"HIP words differ from the independent generator: {name} level {case['level']} {case['op']}"
# CMUXNTT / TRGSW2NTT on the other compiled sets (needs Initialize() only): the second fixture, incl. the chained in-place case
eng.Initialize()
for c, (uniform, extreme), (c1, c0), want in tg.cmux_set_cases():
    idx = ps_index[c["set"]]
    ntt_words = []
    for tgw in (uniform, extreme):
        dtg = api.DeviceBuffer(tgw.size).upload(tgw)
        dntt = api.DeviceBuffer(2 * tgw.size * api.ps_params(idx).key_limbs)     # doubles, once per key limb
        api.ps_trgsw_to_ntt_batch(idx, dtg, dntt, 1)
        ntt_words.append(dntt)
    d1, d0 = api.DeviceBuffer(c1.size).upload(c1), api.DeviceBuffer(c0.size).upload(c0)
    out = api.DeviceBuffer(want.size)
    if c["op"] == "CMUXNTT_CHAINED":
        api.ps_cmux_batch(idx, ntt_words[0], d0, d1, out, 1)        # first = cmux(uniform, operands swapped back)
        api.ps_cmux_batch(idx, ntt_words[1], d1, out, out, 1)       # in place: res = c0
    else:
        api.ps_cmux_batch(idx, ntt_words[c["trgsw"]], d1, d0, out, 1)
    eng.Synchronize()
    assert np.array_equal(out.download(), want), f"HIP CMUXNTT differs from the independent generator: {c['set']} {c['op']} trgsw {c['trgsw']}"
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
