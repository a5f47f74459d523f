"""GPU parity tests of the N = 2048 / 64-bit-torus gate path (BASELINE.json configs[4]).

The reference has no N = 2048 path: parity is against oracle/tfhe_oracle_lvl2.c
(pinned by tests/test_oracle_lvl2.py) word for word, plus decrypt == truth table at the full
4096-gate size.  Bit-exact, no tolerance.
"""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def keys2(oracle, keys):
    return ol.KeysLvl2(oracle, keys, seed=7)


@pytest.fixture(scope="module")
def engine2(engine, keys2):
    engine.lvl2_initialize(keys2.bk, keys2.ksk)
    return engine


def _upload(eng, arr):
    arr = np.ascontiguousarray(arr)
    if arr.dtype == np.uint64:
        arr = arr.view(np.uint32)
    arr = np.ascontiguousarray(arr, dtype=np.uint32)
    return eng.api.DeviceBuffer(arr.size).upload(arr)


def test_params(engine2):
    p = engine2.lvl2_params()
    assert (p.n, p.N, p.l, p.Bgbit, p.t, p.basebit) == (630, 2048, 4, 9, 7, 2)
    assert p.mu == ol.MU2 and p.bk_words == ol.BK2_WORDS and p.ksk_words == ol.KSK2_WORDS


@pytest.fixture(params=["quarter_waves", "half_waves"])
def br2_kernel(request, engine2):
    """both blind-rotate kernels of the ring: four quarter waves per rotation (kernels_lvl2q.hip.h, the default) and
    eight half waves (kernels_lvl2.hip.h); identical words"""
    engine2.api.set_option("lvl2_kernel", 1 if request.param == "quarter_waves" else 0)
    yield request.param
    engine2.api.set_option("lvl2_kernel", -1)


@pytest.mark.parametrize("steps", [0, 1, 2, 3, 33, 630])
def test_blind_rotate_accumulator_words(engine2, keys2, steps, br2_kernel):
    count = 4 if steps == 630 else 8
    rng = np.random.default_rng(200 + steps)
    tl = rng.integers(0, 2**32, size=(count, ol.n + 1), dtype=np.uint64).astype(np.uint32)
    tl[0, :4] = 0                                  # abar = 0 steps
    tl[1, ol.n] = 0                                # bbar = 2N
    tl[2, ol.n] = 0xFFFFFFFF                       # bbar = 1
    tl[3, :8] = 0x7FFFFFFF
    tl[3, ol.n] = 0x80000000                       # bbar = N
    dt = _upload(engine2, tl)
    dacc = engine2.api.DeviceBuffer(count * 2 * ol.N2 * 2)
    engine2.lvl2_blind_rotate_batch(dt, dacc, count, steps)
    got = dacc.download().view(np.uint64).reshape(count, 2 * ol.N2)
    for g in range(count):
        want = keys2.blind_rotate(tl[g], steps)
        assert np.array_equal(got[g], want), f"accumulator of rotation {g} differs after {steps} steps"


def test_keyswitch_words(engine2, keys2):
    count = 12
    rng = np.random.default_rng(9)
    t2 = rng.integers(0, 2**64, size=(count, ol.LVL2_WORDS), dtype=np.uint64)
    t2[0] = 0
    t2[1] = np.uint64(2**64 - 1)
    t2[2, ol.N2] = np.uint64(0x7FFFFFFF80000000)   # rounding of b carries into bit 31
    d2 = _upload(engine2, t2)
    d0 = engine2.api.DeviceBuffer(count * (ol.n + 1))
    engine2.lvl2_keyswitch_batch(d2, d0, count)
    got = d0.download().reshape(count, ol.n + 1)
    for g in range(count):
        assert np.array_equal(got[g], keys2.keyswitch(t2[g])), f"key switch {g} differs"


@pytest.mark.parametrize("count", [129, 203])
def test_keyswitch_words_shared_table_kernel(engine2, keys2, count):
    """keyswitch_kernel over the lvl20 shape (16 ciphertexts per workgroup, table rows through LDS, the 2048 steps of j in at least
    two runs; the default at any count) in three shapes; 129 and 203 leave a ragged last workgroup.  Same words as the oracle
    and as the workgroup-per-ciphertext kernel."""
    rng = np.random.default_rng(count)
    t2 = rng.integers(0, 2**64, size=(count, ol.LVL2_WORDS), dtype=np.uint64)
    t2[0] = 0
    t2[1] = np.uint64(2**64 - 1)
    t2[2, ol.N2] = np.uint64(0x7FFFFFFF80000000)
    t2[3, : ol.N2] = np.uint64(0x8000000000000000)      # every digit at its extreme
    d2 = _upload(engine2, t2)
    d0 = engine2.api.DeviceBuffer(count * (ol.n + 1))
    engine2.api.set_option("ks_wg_threshold", 0)
    try:
        engine2.lvl2_keyswitch_batch(d2, d0, count)
        got = d0.download().reshape(count, ol.n + 1).copy()
        for g in list(range(20)) + list(range(count - 20, count)):
            assert np.array_equal(got[g], keys2.keyswitch(t2[g])), f"key switch {g} differs"
        engine2.api.set_option("ks_wg_threshold", 1 << 20)
        engine2.lvl2_keyswitch_batch(d2, d0, count)
        assert np.array_equal(d0.download().reshape(count, ol.n + 1), got)
        engine2.api.set_option("ks_wg_threshold", 0)
        for per, slices in ((16, 2), (5, 8), (16, 64)):      # (ciphertexts per workgroup, runs of j)
            engine2.api.set_option("ks_per_wg", per)
            engine2.api.set_option("ks_slices", slices)
            d0.upload(np.full(count * (ol.n + 1), 0xDEADBEEF, np.uint32))
            engine2.lvl2_keyswitch_batch(d2, d0, count)
            assert np.array_equal(d0.download().reshape(count, ol.n + 1), got), f"{per} per workgroup, {slices} runs"
    finally:
        for k in ("ks_wg_threshold", "ks_per_wg", "ks_slices"):
            engine2.api.set_option(k, -1)


def test_every_gate_words_and_truth(engine2, keys, keys2, oracle):
    ops = np.arange(len(ol.OPS), dtype=np.int32)
    count = ops.size
    rng = np.random.default_rng(21)
    bits = rng.integers(0, 2, (3, count)).astype(np.uint8)
    cts = [keys.encrypt(bits[i], 0, seed=300 + i) for i in range(3)]
    d = [_upload(engine2, c) for c in cts]
    dout = engine2.api.DeviceBuffer(count * (ol.n + 1))
    engine2.lvl2_gate_batch(ops, dout, d[0], d[1], d[2])
    got = dout.download().reshape(count, ol.n + 1)
    want = keys2.gate_batch(ops, cts[0], cts[1], cts[2])
    for g in range(count):
        assert np.array_equal(got[g], want[g]), f"{ol.OPS[g]} differs from the oracle"
    dec = keys.decrypt(got, 0)
    for g in range(count):
        assert dec[g] == ol.truth(oracle, g, bits[0, g], bits[1, g], bits[2, g]), ol.OPS[g]


def test_4096_nands_decrypt_and_sampled_words(engine2, keys, keys2):
    count = 4096
    rng = np.random.default_rng(33)
    bits = rng.integers(0, 2, (2, count)).astype(np.uint8)
    ca, cb = keys.encrypt(bits[0], 0, seed=401), keys.encrypt(bits[1], 0, seed=402)
    da, db = _upload(engine2, ca), _upload(engine2, cb)
    dout = engine2.api.DeviceBuffer(count * (ol.n + 1))
    engine2.lvl2_gate_batch(ol.OPS.index("NAND"), dout, da, db)
    got = dout.download().reshape(count, ol.n + 1)
    assert np.array_equal(keys.decrypt(got, 0), 1 - (bits[0] & bits[1]))
    pick = rng.choice(count, 8, replace=False)
    want = keys2.gate_batch(ol.OPS.index("NAND"), ca[pick], cb[pick])
    assert np.array_equal(got[pick], want)


def test_errors(engine2):
    buf = engine2.api.DeviceBuffer(ol.n + 1)
    with pytest.raises(engine2.CufheAmdError):
        engine2.lvl2_gate_batch(99, buf, buf, buf)
    with pytest.raises(engine2.CufheAmdError):
        engine2.lvl2_initialize(np.zeros(8, np.uint64), np.zeros(8, np.uint32))


def test_golden_vectors_on_gpu(engine2, keys, keys2):
    """tests/golden/golden_lvl2_v1.json: every op, sha256 of the output words."""
    import hashlib
    import json
    import os
    with open(os.path.join(ol.ROOT, "tests", "golden", "golden_lvl2_v1.json")) as f:
        g = json.load(f)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert sha(keys2.bk) == g["keys_sha256"]["bk"] and sha(keys2.ksk) == g["keys_sha256"]["ksk"], \
        "seeded key generation no longer reproduces tests/golden/golden_lvl2_v1.json (keys_sha256)"
    triples = np.array(g["triples"], np.uint8)
    ins = [keys.encrypt(triples[:, i], 0, seed=7000 + i) for i in range(3)]
    assert [sha(x) for x in ins] == g["inputs_sha256"]
    dins = [_upload(engine2, x) for x in ins]
    dout = engine2.api.DeviceBuffer(len(triples) * (ol.n + 1))
    for op, name in enumerate(ol.OPS):
        engine2.lvl2_gate_batch(op, dout, dins[0], dins[1], dins[2], count=len(triples))
        got = dout.download().reshape(len(triples), -1)
        assert sha(got) == g["ops"][name]["out_sha256"], name
    dacc = engine2.api.DeviceBuffer(2 * ol.N2 * 2)
    engine2.lvl2_blind_rotate_batch(dins[0], dacc, 1, 3)
    assert sha(dacc.download().view(np.uint64)) == g["acc_after_3_steps_sha256"]


def test_lvl0_ring_option_routes_the_level0_api(engine2, keys, keys2):
    """cufhe_amd_set_option("lvl0_ring", 2048): cufhe_amd_gate_batch at level 0 and the recorded
    per-gate API give the words of the lvl2 path."""
    api = engine2.api
    rng = np.random.default_rng(77)
    count = 6
    bits = rng.integers(0, 2, (2, count)).astype(np.uint8)
    ca, cb = keys.encrypt(bits[0], 0, seed=501), keys.encrypt(bits[1], 0, seed=502)
    want = keys2.gate_batch(ol.OPS.index("XOR"), ca, cb)
    da, db = _upload(engine2, ca), _upload(engine2, cb)
    dout = api.DeviceBuffer(count * (ol.n + 1))
    api.set_option("lvl0_ring", 2048)
    try:
        engine2.gate_batch(api.XOR, 0, dout, da, db)
        assert np.array_equal(dout.download().reshape(count, -1), want)
        st = api.Stream()
        st.Create()
        a, b, o = api.Ctxt(0), api.Ctxt(0), api.Ctxt(0)
        a.tlwehost[:] = ca[0]
        b.tlwehost[:] = cb[0]
        api.Xor(o, a, b, st)
        api.Synchronize()
        assert np.array_equal(o.tlwehost, want[0])
        st.Destroy()
    finally:
        api.set_option("lvl0_ring", 1024)
    with pytest.raises(engine2.CufheAmdError):
        api.set_option("lvl0_ring", 512)


def test_blind_rotate_with_extreme_key_words(engine2, keys2, oracle):
    """64-bit key words whose limbs are maximal (-2^63, 2^63 - 1, all-ones, limb boundaries)."""
    rng = np.random.default_rng(19)
    ext = np.array([1 << 63, (1 << 63) - 1, 0, (1 << 64) - 1, (1 << 21), (1 << 21) - 1, (1 << 43), (1 << 43) - 1,
                    0x8000020000200000, 0x7FFFFDFFFFDFFFFF], np.uint64)
    bk = ext[rng.integers(0, ext.size, ol.BK2_WORDS)]
    bk[: 2 * 8 * 2 * ol.N2] = np.uint64(1 << 63)
    ek = oracle.orc2_evalkey_create(bk, keys2.ksk)
    engine2.lvl2_initialize(bk, keys2.ksk)
    try:
        count, steps = 4, 10
        tl = rng.integers(0, 2**32, size=(count, ol.n + 1), dtype=np.uint64).astype(np.uint32)
        dt = _upload(engine2, tl)
        dacc = engine2.api.DeviceBuffer(count * 2 * ol.N2 * 2)
        engine2.lvl2_blind_rotate_batch(dt, dacc, count, steps)
        got = dacc.download().view(np.uint64).reshape(count, 2 * ol.N2)
        for g in range(count):
            want = np.zeros(2 * ol.N2, np.uint64)
            oracle.orc2_blind_rotate(ek, want, np.ascontiguousarray(tl[g]), steps)
            assert np.array_equal(got[g], want), f"rotation {g} differs"
    finally:
        oracle.orc2_evalkey_destroy(ek)
        engine2.lvl2_initialize(keys2.bk, keys2.ksk)


def test_half_wave_layout_is_built_on_first_use(engine2, keys2, keys):
    """The key in the half-transform kernel's layout (495 MB per device) is built when a launch first takes that kernel -- launches of at
    most one rotation per CU, or "lvl2_kernel" 0 -- not by cufhe_amd_lvl2_initialize: a process that only runs batches keeps the half
    gigabyte.  A failing allocation on the way (the "test_fail_alloc" hook) leaves the loaded key usable and leaks nothing."""
    api = engine2.api
    api.Synchronize()
    engine2.lvl2_initialize(keys2.bk, keys2.ksk)             # a fresh load: drops a half layout built by earlier tests
    free_loaded, _ = api.device_mem_info()
    count = 6
    bits = np.array([[0, 1, 0, 1, 1, 0], [0, 0, 1, 1, 0, 1]], np.uint8)
    ins = [keys.encrypt(bits[i], 0, seed=2600 + i) for i in range(2)]
    d0, d1 = _upload(engine2, ins[0]), _upload(engine2, ins[1])
    dout = api.DeviceBuffer(count * (ol.n + 1))
    want = np.asarray(keys2.gate_batch(0, ins[0], ins[1])).reshape(count, -1)
    half_bytes = ol.n * 8 * 2 * ol.N2 * 3 * 8                # n steps x 8 rows x 2 polynomials x N x three limbs, doubles
    api.set_option("lvl2_kernel", 1)
    try:
        api.lvl2_gate_batch(0, dout, d0, d1, count=count)
        api.Synchronize()
        assert np.array_equal(dout.download().reshape(count, -1), want)
        free_q, _ = api.device_mem_info()
        assert free_loaded - free_q < half_bytes // 4, "the quarter-wave kernel must not need the second layout"
        api.set_option("lvl2_kernel", 0)
        api.set_option("test_fail_alloc", 0)                  # the layout's own allocation fails
        with pytest.raises(Exception):
            api.lvl2_gate_batch(0, dout, d0, d1, count=count)
        api.set_option("test_fail_alloc", 1)                  # ... and the staging copy behind it
        with pytest.raises(Exception):
            api.lvl2_gate_batch(0, dout, d0, d1, count=count)
        api.Synchronize()
        free_failed, _ = api.device_mem_info()
        assert abs(free_failed - free_q) < (64 << 20), "a failed build of the half layout leaked device memory"
        api.lvl2_gate_batch(0, dout, d0, d1, count=count)     # builds it now
        api.Synchronize()
        assert np.array_equal(dout.download().reshape(count, -1), want)
        free_h, _ = api.device_mem_info()
        assert free_q - free_h > half_bytes * 9 // 10, "the half-wave kernel's layout was expected to be allocated now"
    finally:
        api.set_option("test_fail_alloc", -1)
        api.set_option("lvl2_kernel", -1)
