"""CPU checks of the N = 2048 / 64-bit-torus oracle (oracle/tfhe_oracle_lvl2.c).

The reference has no N = 2048 path, so this oracle is pinned by
size-independent facts only: its NTT product equals the schoolbook product mod
2^64, and decrypt(gate(...)) equals the reference's truth tables.
"""
import numpy as np
import pytest

import oracle_lib as ol


@pytest.fixture(scope="module")
def keys2(oracle, keys):
    return ol.KeysLvl2(oracle, keys, seed=7)


def test_polymul_ntt_equals_schoolbook_mod_2_64(oracle):
    rng = np.random.default_rng(5)
    for case in range(3):
        a = rng.integers(-256, 256, ol.N2, dtype=np.int32)
        b = rng.integers(0, 1 << 64, ol.N2, dtype=np.uint64)
        if case == 1:                       # extreme magnitudes
            a[:] = -256
            b[:] = np.uint64((1 << 64) - 1)
        r0 = np.zeros(ol.N2, np.uint64)
        r1 = np.zeros(ol.N2, np.uint64)
        oracle.orc2_polymul_schoolbook(r0, a, b)
        oracle.orc2_polymul_ntt(r1, a, b)
        assert np.array_equal(r0, r1)


def test_blind_rotate_keeps_the_message(oracle, keys, keys2):
    # rotation by a fresh encryption of +-mu0 leaves +-mu2 in the constant coefficient
    for bit in (0, 1):
        ct = keys.encrypt([bit], 0, seed=11 + bit)[0]
        acc = keys2.blind_rotate(ct)
        t2 = keys2.sample_extract(acc)
        ph = np.int64(np.uint64(keys2.phase2(t2)))
        err = abs(int(ph) - (ol.MU2 if bit else -ol.MU2))
        assert err < (1 << 57), (bit, ph)
        t0 = keys2.keyswitch(t2)
        assert keys.decrypt(t0, 0)[0] == bit


def test_gates_decrypt_to_truth_table(oracle, keys, keys2):
    ops = [ol.OPS.index(o) for o in ("NAND", "XOR", "ANDYN", "MUX", "NMUX", "NOT", "COPY")]
    rng = np.random.default_rng(3)
    count = 2 * len(ops)
    op_arr = np.array(ops * 2, np.int32)
    bits = rng.integers(0, 2, (3, count)).astype(np.uint8)
    cts = [keys.encrypt(bits[i], 0, seed=100 + i) for i in range(3)]
    out = keys2.gate_batch(op_arr, cts[0], cts[1], cts[2])
    got = keys.decrypt(out, 0)
    want = [ol.truth(oracle, int(op_arr[g]), bits[0, g], bits[1, g], bits[2, g]) for g in range(count)]
    assert list(got) == want
    # the refreshed ciphertexts are fresh lvl0 ciphertexts again: they chain
    out2 = keys2.gate_batch(ol.OPS.index("NAND"), out, cts[1])
    want2 = [int(not (want[g] and bits[1, g])) for g in range(count)]
    assert list(keys.decrypt(out2, 0)) == want2


def test_golden_vectors(oracle, keys, keys2):
    """tests/golden/golden_lvl2_v1.json (made by tests/golden/make_golden_lvl2.py)."""
    import hashlib
    import json
    import os
    with open(os.path.join(ol.ROOT, "tests", "golden", "golden_lvl2_v1.json")) as f:
        g = json.load(f)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert (g["key_seed"], g["key2_seed"]) == (keys.seed, 7)
    for name, arr in (("s0", keys.s0), ("s2", keys2.s2), ("bk", keys2.bk), ("ksk", keys2.ksk)):
        assert sha(arr) == g["keys_sha256"][name], f"key {name} is not reproducible"
    triples = np.array(g["triples"], np.uint8)
    ins = [keys.encrypt(triples[:, i], 0, seed=7000 + i) for i in range(3)]
    assert [sha(x) for x in ins] == g["inputs_sha256"]
    for name in ("NAND", "MUX", "XNOR"):
        out = keys2.gate_batch(ol.OPS.index(name), ins[0], ins[1], ins[2])
        assert sha(out) == g["ops"][name]["out_sha256"], name
        if "out_words_gate0" in g["ops"][name]:
            assert [int(x) for x in out[0]] == g["ops"][name]["out_words_gate0"]
    assert sha(keys2.blind_rotate(ins[0][0], 3)) == g["acc_after_3_steps_sha256"]
