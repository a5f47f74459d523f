"""CPU tests of the oracle (oracle/tfhe_oracle.c): what pins it.

 - the reference's published NTT constants (include/ntt_gpu/ntt_gpuntt.cuh:36-40,
   src/ntt_gpu/ntt_gpuntt.cu:31-32) and its Barrett multiplication against big-int arithmetic;
 - NTT product == schoolbook negacyclic product mod 2^32, the check of
   test/test_polynomial_mult_1024.cu:51-73,209-223 (exact here, the reference allows diff<=2);
 - plaintext truth tables == the reference's own, compiled from /root/reference/test/plain.h
   into oracle/_ref/ (test/plain.h:10-69);
 - decrypt(gate(...)) == truth table for all 13 gate kinds + COPY, both ciphertext levels
   (test/test_gate_gpu.cc:72-84 exercises lvl1, test/test_gate_gpu_multi.cc lvl0);
 - the committed golden vectors (tests/golden/golden_v1.json).
"""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint32).tobytes()).hexdigest()


def test_reference_ntt_constants(oracle):
    p, psi, mu = oracle.orc_ntt_modulus(), oracle.orc_ntt_psi(), oracle.orc_ntt_barrett_mu()
    assert p == 1152921504606877697 == 2**60 + 30721          # ntt_gpuntt.cuh:36
    assert psi == 1689264667710614                              # ntt_gpuntt.cu:32
    assert mu == 9223372036854530040 == (1 << 123) // p         # ntt_gpuntt.cuh:39, bit = 61
    assert pow(psi, 1024, p) == p - 1 and pow(psi, 2048, p) == 1
    assert (p - 1) % 2048 == 0 and (p - 1) % 4096 != 0          # 2-adicity 11: N <= 1024 only
    assert oracle.orc_ntt_n_inverse() == pow(1024, -1, p) == 1151795604700035043


def test_barrett_matches_bigint(oracle):
    p = oracle.orc_ntt_modulus()
    rng = np.random.default_rng(0)
    vals = [0, 1, 2, p - 1, p - 2, p // 2, p // 2 + 1, 2**32, 2**59, 2**60]
    pairs = [(a, b) for a in vals for b in vals]
    pairs += [(int(a), int(b)) for a, b in zip(rng.integers(0, p, 20000, dtype=np.uint64), rng.integers(0, p, 20000, dtype=np.uint64))]
    for a, b in pairs:
        assert oracle.orc_ntt_mulmod(a, b) == a * b % p


def test_ntt_roundtrip_and_spectrum(oracle):
    p = oracle.orc_ntt_modulus()
    rng = np.random.default_rng(1)
    x = rng.integers(0, p, ol.N, dtype=np.uint64)
    y = x.copy()
    oracle.orc_ntt_forward(y)
    # spectrum value at bit-reversed slot br(i) is the evaluation at psi^(2 i + 1)
    psi = oracle.orc_ntt_psi()
    br = lambda v: int(format(v, "010b")[::-1], 2)
    for i in (0, 1, 5, 1023):
        root = pow(psi, 2 * i + 1, p)
        ev = 0
        for c in reversed([int(v) for v in x]):
            ev = (ev * root + c) % p
        assert int(y[br(i)]) == ev
    oracle.orc_ntt_inverse(y)
    assert np.array_equal(x, y)


def test_polymul_ntt_equals_schoolbook(oracle):
    """test/test_polynomial_mult_1024.cu: u32 polynomial x small (<= 18 bit) polynomial."""
    rng = np.random.default_rng(2)
    for trial in range(12):
        bits = [6, 8, 18][trial % 3]
        a = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), ol.N, dtype=np.int32)
        b = rng.integers(0, 2**32, ol.N, dtype=np.uint64).astype(np.uint32)
        if trial == 0:
            a[:] = -32; b[:] = 0x80000000
        if trial == 1:
            a[:] = 31; b[:] = 0xFFFFFFFF
        r1, r2 = np.zeros(ol.N, np.uint32), np.zeros(ol.N, np.uint32)
        oracle.orc_polymul_schoolbook(r1, a, b)
        oracle.orc_polymul_ntt(r2, a, b)
        assert np.array_equal(r1, r2)


REF_NAMES = ol.REF_NAMES


def test_truth_tables_match_reference_plain_h(oracle):
    """oracle/_ref/libplain_ref.so is /root/reference/test/plain.h compiled as is: the one pin the reference itself
    holds for this path (test/test_util.h:75-94 checks decrypt == these functions).  A missing library is a FAILURE
    wherever the reference tree is mounted (the build step must have produced it) and wherever a built copy should have
    travelled with the repo; it may only be absent on a machine that has neither."""
    if not os.path.exists(ol.REF_LIB):
        assert not os.path.exists("/root/reference/test/plain.h"), \
            "oracle/_ref/libplain_ref.so is missing although /root/reference is mounted: run `make -C oracle all`"
        pytest.skip("neither /root/reference nor a built oracle/_ref on this machine")
    for name in REF_NAMES:
        op = ol.OPS.index(name)
        for a in (0, 1):
            for b in (0, 1):
                for c in (0, 1):
                    assert ol.ref_truth(name, a, b, c) == oracle.orc_truth(op, a, b, c), (name, a, b, c)
                    assert ol.truth(oracle, op, a, b, c) == ol.ref_truth(name, a, b, c)      # what every GPU truth check uses
    # NOR is in the reference's gate set (src/bootstrap_gpu.cu:433-440) but not in plain.h
    assert ol.ref_truth("NOR", 0, 0) is None
    assert [ol.truth(oracle, ol.OPS.index("NOR"), a, b) for a in (0, 1) for b in (0, 1)] == [1, 0, 0, 0]


def test_encrypt_decrypt_and_noise(keys):
    rng = np.random.default_rng(4)
    for level in (0, 1):
        bits = rng.integers(0, 2, 256).astype(np.uint8)
        cts = keys.encrypt(bits, level, seed=31 + level)
        assert np.array_equal(keys.decrypt(cts, level), bits)
        # phase = +-mu + e with tiny e
        key = keys.key(level)
        ph = (cts[:, -1] - (cts[:, :-1] * key).sum(axis=1, dtype=np.uint64).astype(np.uint32)).astype(np.int32)
        err = ph.astype(np.int64) - np.where(bits == 1, ol.MU, -ol.MU)
        assert np.abs(err).max() < ol.MU // 8


@pytest.mark.parametrize("level", [0, 1])
def test_every_gate_decrypts_to_truth_table(keys, oracle, level):
    combos = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], np.uint8)
    ins = [keys.encrypt(combos[:, i], level, seed=900 + 10 * level + i) for i in range(3)]
    for op in range(14):
        out = keys.gate_batch(op, level, ins[0], ins[1], ins[2])
        assert list(keys.decrypt(out, level)) == [ol.truth(oracle, op, *c) for c in combos], ol.OPS[op]


def test_pieces_compose_to_gate(keys, oracle):
    """blind rotate -> sample extract -> key switch == orc_gate (lvl0 NAND)."""
    ins = [keys.encrypt([1], 0, seed=61), keys.encrypt([0], 0, seed=62)]
    c = (-ins[0][0].astype(np.int64) - ins[1][0].astype(np.int64)).astype(np.uint32)
    c[ol.n] = (int(c[ol.n]) + ol.MU) % 2**32
    acc = np.zeros(2 * ol.N, np.uint32)
    oracle.orc_blind_rotate(keys.ek, acc, np.ascontiguousarray(c), -1)
    t1 = np.zeros(ol.N + 1, np.uint32)
    oracle.orc_sample_extract0(t1, acc)
    t0 = np.zeros(ol.n + 1, np.uint32)
    oracle.orc_keyswitch(keys.ek, t0, t1)
    want = keys.gate_batch(0, 0, ins[0], ins[1])[0]
    assert np.array_equal(t0, want) and keys.decrypt(t0, 0)[0] == 1
    assert keys.decrypt(t1, 1)[0] == 1      # the extracted lvl1 TLWE already decrypts


def test_golden_vectors(keys, oracle):
    with open(os.path.join(ol.ROOT, "tests", "golden", "golden_v1.json")) as f:
        g = json.load(f)
    assert g["key_seed"] == keys.seed
    for name in ("s0", "s1", "bk", "ksk"):
        assert sha(getattr(keys, name)) == g["keys_sha256"][name], f"key {name} is not reproducible"
    triples = np.array(g["triples"], np.uint8)
    for level in (0, 1):
        ins = [keys.encrypt(triples[:, i], level, seed=5000 + 100 * level + i) for i in range(3)]
        lv = g["levels"][str(level)]
        assert [sha(x) for x in ins] == lv["inputs_sha256"]
        for op, name in enumerate(ol.OPS):
            out = keys.gate_batch(op, level, ins[0], ins[1], ins[2])
            assert sha(out) == lv["ops"][name]["out_sha256"], (name, level)
            assert [int(x) for x in keys.decrypt(out, level)] == lv["ops"][name]["decrypt"]
            if "out_words_gate0" in lv["ops"][name]:
                assert [int(x) for x in out[0]] == lv["ops"][name]["out_words_gate0"]


def test_fast_cpu_baseline_words_match_oracle(keys, oracle):
    """oracle/cpu_fast.c (the optimised CPU gate bench.py times as `cpu_baseline`) computes the oracle's
    words: every two-input op, inputs on both sides of the rotation wrap-around, ragged block sizes."""
    rng = np.random.default_rng(77)
    count = 21                                   # one full block of 16 and a ragged one
    bits = rng.integers(0, 2, size=(2, count)).astype(np.uint8)
    ins = [keys.encrypt(bits[i], 0, seed=770 + i) for i in range(2)]
    ins[0][0, :] = 0                             # abar = 0 steps, bbar from the gate offset alone
    ins[1][1, ol.n] = 0xFFFFFFFF
    ops = np.array([g % 10 for g in range(count)], np.int32)
    fek = oracle.fast_evalkey_create(keys.bk, keys.ksk)
    try:
        out = np.zeros(count * (ol.n + 1), np.uint32)
        assert oracle.fast_gate_batch(fek, ops, 1, count, out, ins[0].ravel(), ins[1].ravel(), 4) == 0
        want = keys.gate_batch(ops, 0, ins[0], ins[1])
        assert np.array_equal(out.reshape(count, -1), want)
        assert oracle.fast_gate_batch(fek, np.array([10], np.int32), 0, 1, out, ins[0].ravel(), ins[1].ravel(), 1) == -1   # MUX: not a two-input op
    finally:
        oracle.fast_evalkey_destroy(fek)


def test_small_modulus_oracle():
    """oracle/liboracle_smallmod.so: the restatement of the reference built with -DUSE_SMALL_NTT_MODULUS (CMakeLists.txt:12;
    include/ntt_gpu/ntt_small_modulus.cuh).  Pinned here: its constants (P = 625 * 2^20 + 1, :36-68; psi a primitive 2048-th root
    found as src/ntt_gpu/ntt_small_modulus.cu:71-110 finds it), the two modulus switches against their formulas in Python integers
    (:147-177), and -- the mode is approximate by design -- that every gate still decrypts to the reference's truth table on both
    ciphertext levels.  Word level: tests/test_golden_independent.py (schoolbook mod P)."""
    L = ol.load_set("smallmod")
    for f in ("orc_ntt_modulus", "orc_ntt_psi", "orc_ntt_n_inverse"):
        getattr(L, f).restype = ctypes.c_uint64
    L.orc_smallmod_from_torus.restype = ctypes.c_uint32
    L.orc_smallmod_from_torus.argtypes = [ctypes.c_uint32]
    L.orc_smallmod_to_torus.restype = ctypes.c_uint32
    L.orc_smallmod_to_torus.argtypes = [ctypes.c_int32]
    P = 625 * 2**20 + 1
    assert L.orc_ntt_modulus() == P == 655360001 and L.orc_smallmod_modulus() == P
    psi = L.orc_ntt_psi()
    assert pow(psi, 1024, P) == P - 1 and pow(psi, 2048, P) == 1
    g = next(x for x in range(3, 1000) if pow(x, (P - 1) // 2, P) != 1 and pow(x, (P - 1) // 5, P) != 1)
    assert psi == pow(g, (P - 1) // 2048, P)
    assert L.orc_ntt_n_inverse() * 1024 % P == 1
    rng = np.random.default_rng(9)
    inv = 2**63 // P
    for a in [0, 1, 2, 3, 0x7FFFFFFF, 0x80000000, 0x80000001, 0xFFFFFFFE, 0xFFFFFFFF] + [int(x) for x in rng.integers(0, 2**32, 200)]:
        assert L.orc_smallmod_from_torus(a) == (a * P + 2**31) >> 32
    assert L.orc_smallmod_from_torus(0xFFFFFFFF) == P                 # the one unreduced value: 0 mod P
    for r in [0, 1, P // 2, P // 2 + 1, P - 1] + [int(x) for x in rng.integers(0, P, 200)]:
        signed = r - P if r > P // 2 else r                            # the centred value the kernels hand over
        assert L.orc_smallmod_to_torus(signed) == ((r * inv + 2**30) >> 31) & 0xFFFFFFFF
        assert abs(((L.orc_smallmod_to_torus(signed) - round(r * 2**32 / P) + 2**31) % 2**32) - 2**31) <= 1
    K = ol.Keys(L, seed=7)
    combos = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], np.uint8)
    for level in (0, 1):
        ins = [K.encrypt(combos[:, i], level, seed=700 + 10 * level + i) for i in range(3)]
        for op in range(14):
            out = K.gate_batch(op, level, ins[0], ins[1], ins[2])
            assert list(K.decrypt(out, level)) == [ol.truth(L, op, *c) for c in combos], (ol.OPS[op], level)
    # and the words are NOT the exact path's (the same keys through liboracle.so): the mode is a different function
    E = ol.load()
    KE = ol.Keys(E, seed=7)
    assert np.array_equal(KE.bk, K.bk)
    ins = [K.encrypt(combos[:, i], 0, seed=700 + i) for i in range(2)]
    a, b = K.gate_batch(0, 0, ins[0], ins[1]), KE.gate_batch(0, 0, ins[0], ins[1])
    assert not np.array_equal(a, b) and np.abs((a.astype(np.int64) - b.astype(np.int64) + 2**31) % 2**32 - 2**31)[:, :-1].max() > 0
