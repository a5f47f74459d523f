"""Other parameter sets on the GPU: the generic kernels of cufhe_amd/csrc/kernels_ps.hip.h
against the CPU oracle compiled for the same set (oracle/liboracle_<set>.so), word for word.

  default   n = 630, N = 1024, k = 1, l = 3, Bg = 2^6     also == the hand-scheduled kernels' words
  k2n512    n = 630, N = 512,  k = 2, l = 3, Bg = 2^6     the k > 1 handling of src/bootstrap_gpu.cu:402-421 and the
                                                          512-point transform of include/ntt_gpu/ntt_gpuntt.cuh:283-329
  cggi16    n = 500, N = 1024, k = 1, l = 2, Bg = 2^10    the external product leaves the exact range of the FP64 prime:
                                                          two 16-bit key limbs (the fallback of kernels_lvl2.hip.h)
"""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=ol.SETS)
def pset(request, engine):
    name = request.param
    L = ol.load_set(name)
    K = ol.Keys(L, seed=5)
    idx = engine.api.ps_index(name)
    p = engine.api.ps_params(idx)
    _, want = ol.set_params(L)
    assert (p.n, p.N, p.k, p.l, p.Bgbit, p.t, p.basebit) == tuple(want[k] for k in ("n", "N", "k", "l", "Bgbit", "t", "basebit"))
    assert (p.bk_words, p.ksk_words, p.lvl0_words, p.lvl1_words) == (K.bk_words, K.ksk_words, K.words[0], K.words[1])
    engine.api.ps_initialize(idx, K.bk, K.ksk)
    return name, idx, L, K


@pytest.fixture(autouse=True, params=["wg", "batch"])
def ps_kernel(request, engine):
    """Both launch shapes of the blind rotation: a workgroup per rotation (small launches) and a wave per rotation with
    eight rotations per workgroup sharing the key rows (above 1024 to 1536 rotations, by set); forced here for every count."""
    engine.api.set_option("ps_batch_threshold", 1 << 30 if request.param == "wg" else 1)
    yield request.param
    engine.api.set_option("ps_batch_threshold", -1)


def _up(eng, a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return eng.api.DeviceBuffer(a.size).upload(a)


@pytest.mark.parametrize("steps", [0, 1, 2, 3, 41, -1])
def test_accumulator_words(engine, pset, steps):
    name, idx, L, K = pset
    count = 4 if steps == -1 else 13       # 13: a second workgroup with idle waves in the batch shape (8 or 12 rotations per workgroup)
    rng = np.random.default_rng(50 + steps)
    tl = rng.integers(0, 2**32, size=(count, K.n + 1), dtype=np.uint64).astype(np.uint32)
    tl[0, :4] = 0                      # abar = 0 steps
    tl[1, K.n] = 0                     # bbar = 2N
    tl[2, K.n] = 0xFFFFFFFF            # bbar = 1
    tl[3, :8] = 0x7FFFFFFF
    tl[3, K.n] = 0x80000000            # bbar = N
    dacc = engine.api.DeviceBuffer(count * (K.k + 1) * K.N)
    engine.api.ps_blind_rotate_batch(idx, _up(engine, tl), dacc, count, steps)
    got = dacc.download().reshape(count, -1)
    for g in range(count):
        assert np.array_equal(got[g], K.blind_rotate(tl[g], steps)), f"{name}: accumulator {g} differs after {steps} steps"


def test_extreme_key_words(engine, pset):
    """Key words of maximal magnitude: the sums (per key limb) run closest to the p/2 bound."""
    name, idx, L, K = pset
    rng = np.random.default_rng(3)
    ext = np.array([0x80000000, 0x7FFFFFFF, 0, 0xFFFFFFFF, 0x80000001, 0x7FFF8000, 0x8000], np.uint32)
    bk = ext[rng.integers(0, ext.size, K.bk_words)]
    step = K.bk_words // K.n
    bk[: 2 * step] = 0x80000000
    ek = L.orc_evalkey_create(bk, K.ksk)
    engine.api.ps_initialize(idx, bk, K.ksk)
    try:
        count, steps = 4, 9
        tl = rng.integers(0, 2**32, size=(count, K.n + 1), dtype=np.uint64).astype(np.uint32)
        dacc = engine.api.DeviceBuffer(count * (K.k + 1) * K.N)
        engine.api.ps_blind_rotate_batch(idx, _up(engine, tl), dacc, count, steps)
        got = dacc.download().reshape(count, -1)
        for g in range(count):
            want = np.zeros((K.k + 1) * K.N, np.uint32)
            L.orc_blind_rotate(ek, want, np.ascontiguousarray(tl[g]), steps)
            assert np.array_equal(got[g], want), f"{name}: rotation {g}"
    finally:
        L.orc_evalkey_destroy(ek)
        engine.api.ps_initialize(idx, K.bk, K.ksk)


def test_keyswitch_words(engine, pset):
    name, idx, L, K = pset
    count = 23
    rng = np.random.default_rng(6)
    t1 = rng.integers(0, 2**32, size=(count, K.words[1]), dtype=np.uint64).astype(np.uint32)
    t1[0] = 0
    t1[1] = 0xFFFFFFFF
    d0 = engine.api.DeviceBuffer(count * K.words[0])
    # -1: keyswitch_kernel over the set's shape, launch shape by the rule; 1 << 20: a workgroup per ciphertext; then forced shapes
    # (ciphertexts per workgroup, runs of j)
    for thr, per, slices in ((-1, -1, -1), (1 << 20, -1, -1), (0, 16, 1), (0, 7, 4), (0, 16, 64)):
        engine.api.set_option("ks_wg_threshold", thr)
        engine.api.set_option("ks_per_wg", per)
        engine.api.set_option("ks_slices", slices)
        try:
            d0.upload(np.full(count * K.words[0], 0xDEADBEEF, np.uint32))
            engine.api.ps_keyswitch_batch(idx, _up(engine, t1), d0, count)
        finally:
            for k in ("ks_wg_threshold", "ks_per_wg", "ks_slices"):
                engine.api.set_option(k, -1)
        got = d0.download().reshape(count, -1)
        for g in range(count):
            assert np.array_equal(got[g], K.keyswitch(t1[g])), f"{name}: key switch {g} (ks_wg_threshold {thr}, {per} per workgroup, {slices} runs)"


@pytest.mark.parametrize("level", [0, 1])
def test_every_gate_words_and_truth_table(engine, pset, keys, level):
    """level 0: __HomGate__ br -> iks on n + 1 words (src/bootstrap_gpu.cu:402-421, Mux :515-588); level 1: the other order on
    k N + 1 words, key switch of the linear combination first (:383-400, Mux :706-780) -- every op, every set, both launch shapes"""
    name, idx, L, K = pset
    combos = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], np.uint8)
    count = len(combos)
    ins = [K.encrypt(combos[:, i], level, seed=900 + i) for i in range(3)]
    dins = [_up(engine, x) for x in ins]
    dout = engine.api.DeviceBuffer(count * K.words[level])
    for op in range(14):
        engine.api.ps_gate_batch(idx, op, dout, dins[0], dins[1], dins[2], count=count, level=level)
        got = dout.download().reshape(count, -1)
        assert np.array_equal(got, K.gate_batch(op, level, ins[0], ins[1], ins[2])), f"{name} {ol.OPS[op]} level {level}: words differ from the oracle"
        assert list(K.decrypt(got, level)) == [ol.truth(L, op, *map(int, c)) for c in combos], f"{name} {ol.OPS[op]} level {level}: decrypt != truth table"
    if name == "default" and level == 0:
        # the generic kernels and the hand-scheduled ones are two implementations of one function
        engine.Initialize(K.bk, K.ksk)
        try:
            for op in (0, 5, 10):
                engine.api.ps_gate_batch(idx, op, dout, dins[0], dins[1], dins[2], count=count)
                a = dout.download().copy()
                engine.gate_batch(op, 0, dout, dins[0], dins[1], dins[2], count=count)
                assert np.array_equal(a, dout.download())
        finally:
            engine.Initialize(keys.bk, keys.ksk)


def test_trlwe_level_operations_over_a_parameter_set(engine, pset):
    """GateBootstrappingTLWE2TRLWElvl01NTT, Refresh and SampleExtractAndKeySwitch (src/cufhe_gates_gpu.cu:86-146) recorded through the
    per-gate API on another set -- k + 1 = 3 polynomials of 512 for `k2n512`, the small NTT modulus for `smallmod` (whose build in the
    reference keeps exactly these three and drops CMUXNTT, :68-86) -- word for word against the set's oracle, chained without a
    synchronisation in between."""
    name, idx, L, K = pset
    api = engine.api
    api.set_option("param_set", idx)
    try:
        count = 6
        trlwe_words = (K.k + 1) * K.N
        assert api.Trlwe().trlwehost.size == trlwe_words
        bits = np.array([g & 1 for g in range(count)], np.uint8)
        enc = K.encrypt(bits, 0, seed=4100)
        ins = [api.Ctxt(0) for _ in range(count)]
        t, r, o = ([api.Trlwe() for _ in range(count)] for _ in range(3))
        outs = [api.Ctxt(0) for _ in range(count)]
        sts = [api.Stream() for _ in range(2)]
        for s in sts:
            s.Create()
        for g in range(count):
            ins[g].tlwehost[:] = enc[g]
            api.GateBootstrappingTLWE2TRLWElvl01NTT(t[g], ins[g], sts[g % 2])
            api.Refresh(r[g], t[g], sts[g % 2])
            api.SampleExtractAndKeySwitch(outs[g], r[g], sts[g % 2])
        api.Synchronize()

        def extract(acc):
            tl = np.zeros(K.words[1], np.uint32)
            L.orc_sample_extract0(tl, np.ascontiguousarray(acc))
            return tl
        for g in range(count):
            want_t = K.blind_rotate(enc[g])
            assert np.array_equal(t[g].trlwehost, want_t), f"{name}: bootstrap to a TRLWE {g}"
            want_r = K.blind_rotate(K.keyswitch(extract(want_t)))
            assert np.array_equal(r[g].trlwehost, want_r), f"{name}: Refresh {g}"
            want_o = K.keyswitch(extract(want_r))
            assert np.array_equal(outs[g].tlwehost, want_o), f"{name}: SampleExtractAndKeySwitch {g}"
            assert K.decrypt(want_o[None, :], 0)[0] == bits[g] and K.decrypt(extract(want_r)[None, :], 1)[0] == bits[g]
        for s in sts:
            s.Destroy()
    finally:
        api.set_option("param_set", -1)
    assert api.Trlwe().trlwehost.size == 2 * ol.N
    # the device-resident batch form of the same three (cufhe_amd_ps_trlwe_op_batch), without "param_set"
    d_in = _up(engine, enc)
    d_t, d_r = api.DeviceBuffer(count * trlwe_words), api.DeviceBuffer(count * trlwe_words)
    d_o = api.DeviceBuffer(count * K.words[0])
    api.ps_trlwe_op_batch(idx, api.TL_BOOTSTRAP, d_t, d_in, count)
    api.ps_trlwe_op_batch(idx, api.TL_REFRESH, d_r, d_t, count)
    api.ps_trlwe_op_batch(idx, api.TL_SEIKS, d_o, d_r, count)
    assert np.array_equal(d_t.download().reshape(count, -1), np.stack([x.trlwehost for x in t]))
    assert np.array_equal(d_r.download().reshape(count, -1), np.stack([x.trlwehost for x in r]))
    assert np.array_equal(d_o.download().reshape(count, -1), np.stack([x.tlwehost for x in outs]))
    with pytest.raises(Exception):
        api.ps_trlwe_op_batch(idx, 103, d_r, d_t, 1)          # CMUXNTT takes four operands: cufhe_amd_ps_cmux_batch


def test_cmux_and_trgsw2ntt_over_a_parameter_set(engine, pset):
    """TRGSW2NTT + CMUXNTT (src/bootstrap_gpu.cu:75-94,197-285: templates over the lvl1param the build selected) on every compiled set but
    the small-modulus one (whose build in the reference has none, src/cufhe_gates_gpu.cu:68-86): res = c0 + trgsw [x] (c1 - c0) word for
    word against the set's oracle -- uniform operands, a TRGSW of extreme words (cggi16: the sums that need both key limbs), a row of the
    bootstrapping key (a real TRGSW encryption of a key bit), and in place on c0 and on c1."""
    name, idx, L, K = pset
    api = engine.api
    l = ol.set_params(L)[1]["l"]
    trlwe_words, trgsw_words = (K.k + 1) * K.N, (K.k + 1) * l * (K.k + 1) * K.N
    count = 7
    rng = np.random.default_rng(77)
    tg = rng.integers(0, 2**32, size=(count, trgsw_words), dtype=np.uint64).astype(np.uint32)
    ext = np.array((0x80000000, 0x7FFFFFFF, 0, 0xFFFFFFFF, 0x80000001), np.uint32)
    tg[1] = ext[rng.integers(0, ext.size, trgsw_words)]
    tg[2] = 0x80000000
    tg[3] = np.asarray(K.bk, np.uint32).reshape(K.n, trgsw_words)[5]
    c1 = rng.integers(0, 2**32, size=(count, trlwe_words), dtype=np.uint64).astype(np.uint32)
    c0 = rng.integers(0, 2**32, size=(count, trlwe_words), dtype=np.uint64).astype(np.uint32)
    c1[4, :64] = c0[4, :64]                      # difference 0: the digits of the bare offset
    c1[5] = c0[5] + np.uint32(0x7FFFFFFF)
    d_tg = _up(engine, tg)
    limbs = api.ps_params(idx).key_limbs
    d_ntt = api.DeviceBuffer(count * 2 * trgsw_words * limbs)
    if name == "smallmod":
        with pytest.raises(Exception, match="small-modulus"):
            api.ps_trgsw_to_ntt_batch(idx, d_tg, d_ntt, count)
        with pytest.raises(Exception, match="small-modulus"):
            api.ps_cmux_batch(idx, d_ntt, _up(engine, c1), _up(engine, c0), api.DeviceBuffer(count * trlwe_words), count)
        return
    api.ps_trgsw_to_ntt_batch(idx, d_tg, d_ntt, count)
    d1, d0, dres = _up(engine, c1), _up(engine, c0), api.DeviceBuffer(count * trlwe_words)
    api.ps_cmux_batch(idx, d_ntt, d1, d0, dres, count)
    got = dres.download().reshape(count, -1)
    want = np.zeros_like(got)
    for g in range(count):
        L.orc_cmux(want[g], np.ascontiguousarray(tg[g]), np.ascontiguousarray(c1[g]), np.ascontiguousarray(c0[g]))
        assert np.array_equal(got[g], want[g]), f"{name}: CMUXNTT {g}"
    api.ps_cmux_batch(idx, d_ntt, d1, d0, d0, count)       # in place on c0 (the reference's usual call, test/test_cmux.cc)
    assert np.array_equal(d0.download().reshape(count, -1), want), f"{name}: CMUXNTT in place on c0"
    d0 = _up(engine, c0)
    api.ps_cmux_batch(idx, d_ntt, d1, d0, d1, count)       # ... and on c1
    assert np.array_equal(d1.download().reshape(count, -1), want), f"{name}: CMUXNTT in place on c1"
    # the selector works: with TRGSW(s) of a key bit s the result decrypts to c1's message if s else c0's
    bits = np.array([1, 0], np.uint8)
    s0 = np.asarray(K.s0)
    for i in (int(np.argmax(s0 == 1)), int(np.argmax(s0 == 0))):
        row = np.ascontiguousarray(np.asarray(K.bk, np.uint32).reshape(K.n, trgsw_words)[i])
        t = [K.blind_rotate(K.encrypt(bits[j:j + 1], 0, seed=900 + j)[0]) for j in range(2)]      # TRLWEs of mu * (+-1) at coefficient 0
        d_row, d_rn = _up(engine, row), api.DeviceBuffer(2 * trgsw_words * limbs)
        api.ps_trgsw_to_ntt_batch(idx, d_row, d_rn, 1)
        dr = api.DeviceBuffer(trlwe_words)
        api.ps_cmux_batch(idx, d_rn, _up(engine, t[0]), _up(engine, t[1]), dr, 1)
        tl = np.zeros(K.words[1], np.uint32)
        L.orc_sample_extract0(tl, dr.download())
        assert K.decrypt(tl[None, :], 1)[0] == (bits[0] if s0[i] else bits[1]), f"{name}: CMUXNTT selects the wrong operand for key bit {int(s0[i])}"


def test_mixed_batch(engine, pset):
    name, idx, L, K = pset
    count = 48
    rng = np.random.default_rng(12)
    bits = rng.integers(0, 2, size=(3, count)).astype(np.uint8)
    ins = [K.encrypt(bits[i], 0, seed=1300 + i) for i in range(3)]
    ops = np.array([[3, 4, 5, 0, 10, 12, 13, 11][g % 8] for g in range(count)], np.int32)
    dins = [_up(engine, x) for x in ins]
    engine.api.ps_gate_batch(idx, ops, dins[0], dins[0], dins[1], dins[2], count=count)     # out aliases in0
    got = dins[0].download().reshape(count, -1)
    assert np.array_equal(got, K.gate_batch(ops, 0, ins[0], ins[1], ins[2]))
    assert list(K.decrypt(got, 0)) == [ol.truth(L, int(ops[g]), int(bits[0, g]), int(bits[1, g]), int(bits[2, g])) for g in range(count)]


@pytest.mark.parametrize("level", [0, 1])
def test_per_gate_api_over_a_parameter_set(engine, pset, level):
    """cufhe_amd_set_option("param_set"): the Stream / Ctxt / Nand ... surface and its scheduler over another set, both ciphertext
    levels -- the 80-bit set included, whose lvl0 ciphertexts have 501 words (Ctxt takes its size from the active set; the device
    slots are carved for the largest compiled size)."""
    name, idx, L, K = pset
    api = engine.api
    api.set_option("param_set", idx)
    try:
        assert [api.Ctxt(lv).tlwehost.size for lv in (0, 1)] == [K.words[0], K.words[1]]
        count = 24
        rng = np.random.default_rng(31)
        bits = rng.integers(0, 2, size=(3, count)).astype(np.uint8)
        enc = [K.encrypt(bits[i], level, seed=3100 + i) for i in range(3)]
        cts = [[api.Ctxt(level) for _ in range(count)] for _ in range(3)]
        for i in range(3):
            for g in range(count):
                cts[i][g].tlwehost[:] = enc[i][g]
        outs = [api.Ctxt(level) for _ in range(count)]
        sts = [api.Stream() for _ in range(4)]
        for s in sts:
            s.Create()
        for g in range(count):
            (api.Nand, api.Xor, api.Mux)[g % 3](outs[g], cts[0][g], cts[1][g], *((cts[2][g],) if g % 3 == 2 else ()), sts[g % 4])
        api.Synchronize()
        ops = np.array([[0, 5, 10][g % 3] for g in range(count)], np.int32)
        got = np.stack([o.tlwehost for o in outs])
        assert np.array_equal(got, K.gate_batch(ops, level, enc[0], enc[1], enc[2]))
        assert list(K.decrypt(got, level)) == [ol.truth(L, int(ops[g]), int(bits[0, g]), int(bits[1, g]), int(bits[2, g])) for g in range(count)]
        for s in sts:
            s.Destroy()
    finally:
        api.set_option("param_set", -1)
    assert api.Ctxt(0).tlwehost.size == ol.n + 1
    if K.words[level] != ol.LVL_WORDS[level]:
        # a ciphertext keeps the host buffer of the set it was created under: using it under another set is refused, not copied past its end
        st = api.Stream()
        st.Create()
        with pytest.raises(Exception, match="another parameter set"):
            api.Nand(api.Ctxt(level), cts[0][0], cts[1][0], st)
        with pytest.raises(Exception, match="another parameter set"):
            api.CtxtCopyH2D(cts[0][0], st)
        st.Destroy()
