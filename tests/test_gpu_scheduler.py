"""The per-gate API (Stream / Ctxt / Nand ... NMux, g-gates, explicit copies) on the GPU, checked
WORD FOR WORD against the CPU oracle -- not only by decryption -- and by the scheduler's own launch
counters: dependent gates must be cut into dependence levels, not into 1-gate launches.

Reference programs: test/test_api_gpu.cu:140-159 (chained in-place gates), test/test_gate_gpu.cc:36-91
and test/test_util.h:29-94 (every gate over many streams), test/test_intensive.cc:21-128 (polling),
include/cufhe_gpu.cuh:282-313 (g-gates).
"""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
O = ol.OPS.index


def _ctxts(api, keys, bits, level, seed):
    enc = keys.encrypt(bits, level, seed=seed)
    cts = []
    for row in enc:
        c = api.Ctxt(level)
        c.tlwehost[:] = row
        cts.append(c)
    return cts, enc


def _host(cts):
    return np.stack([c.tlwehost.copy() for c in cts])


@pytest.mark.parametrize("level", [0, 1])
def test_enqueue_gate_words(engine, keys, level):
    """All 14 ops through cufhe_amd_enqueue_gate on 8 streams: delivered words == oracle."""
    api = engine.api
    count = 56
    rng = np.random.default_rng(60 + level)
    bits = rng.integers(0, 2, size=(3, count)).astype(np.uint8)
    ins = [_ctxts(api, keys, bits[i], level, 6100 + 10 * level + i) for i in range(3)]
    outs = [api.Ctxt(level) for _ in range(count)]
    sts = [api.Stream() for _ in range(8)]
    for s in sts:
        s.Create()
    fns2 = {"NAND": api.Nand, "NOR": api.Nor, "XNOR": api.Xnor, "AND": api.And, "OR": api.Or, "XOR": api.Xor,
            "ANDNY": api.AndNY, "ANDYN": api.AndYN, "ORNY": api.OrNY, "ORYN": api.OrYN}
    ops = np.array([g % 14 for g in range(count)], np.int32)
    api.sched_stats(reset=True)
    for g in range(count):
        name, st = ol.OPS[ops[g]], sts[g % 8]
        a, b, c = ins[0][0][g], ins[1][0][g], ins[2][0][g]
        if name in fns2:
            fns2[name](outs[g], a, b, st)
        elif name == "MUX":
            api.Mux(outs[g], a, b, c, st)
        elif name == "NMUX":
            api.NMux(outs[g], a, b, c, st)
        elif name == "NOT":
            api.Not(outs[g], a, st)
        else:
            api.Copy(outs[g], a, st)
    api.Synchronize()
    stats = api.sched_stats()
    assert stats.gates == count and stats.launch_sequences == 1, (stats.gates, stats.launch_sequences)
    want = keys.gate_batch(ops, level, ins[0][1], ins[1][1], ins[2][1])
    assert np.array_equal(_host(outs), want), "words delivered through enqueue_gate differ from the oracle"
    for s in sts:
        s.Destroy()


@pytest.mark.parametrize("zero_copy", [1, 0])
def test_staging_paths_and_flush_timeline(engine, keys, zero_copy):
    """Both ways a flush moves its ciphertexts -- the scatter / gather kernels working on the pinned block itself ("sched_zero_copy", the
    default; the first level's inputs go chunk by chunk behind the gather) and the round-3 path through device staging copies -- deliver the
    oracle's words, and every flush leaves a consistent record in cufhe_amd_sched_get_trace (include/cufhe_amd.h)."""
    api = engine.api
    api.set_option("sched_zero_copy", zero_copy)
    try:
        count = 1100                   # three chunks of 512 inputs in the first level, a ragged last one
        rng = np.random.default_rng(901 + zero_copy)
        bits = rng.integers(0, 2, size=(2, count)).astype(np.uint8)
        (a, ea), (b, eb) = (_ctxts(api, keys, bits[i], 0, 9100 + i) for i in range(2))
        outs = [api.Ctxt(0) for _ in range(count)]
        outs2 = [api.Ctxt(0) for _ in range(count)]
        sts = [api.Stream() for _ in range(16)]
        for s in sts:
            s.Create()
        api.Synchronize()
        api.sched_trace(clear=True)
        api.profile_enable(True)
        for g in range(count):
            api.Xor(outs[g], a[g], b[g], sts[g % 16])
        for g in range(count):         # a second level: reads results still on their way to tlwehost
            api.Nand(outs2[g], outs[g], b[g], sts[g % 16])
        api.Synchronize()
        api.profile_enable(False)
        api.profile_get(reset=True)
        w1 = keys.gate_batch(O("XOR"), 0, ea, eb)
        assert np.array_equal(_host(outs), w1)
        assert np.array_equal(_host(outs2), keys.gate_batch(O("NAND"), 0, w1, eb))
        tr = api.sched_trace()
        assert tr and sum(t["gates"] for t in tr) == 2 * count
        assert sum(t["in_bytes"] for t in tr) == 2 * count * (ol.n + 1) * 4           # a, b uploaded once; outs never re-uploaded
        assert sum(t["out_bytes"] for t in tr) == 2 * count * (ol.n + 1) * 4
        for t in tr:
            assert t["t_queued"] <= t["t_launch_begin"] <= t["t_gather_end"] <= t["t_submit_end"] <= t["t_done_seen"] <= t["t_delivered"], t
            assert t["dev_h2d_ms"] >= 0 and t["dev_body_ms"] > 0 and t["dev_d2h_ms"] >= 0, t
        for s in sts:
            s.Destroy()
    finally:
        api.set_option("sched_zero_copy", 1)


def test_chained_program_levels_and_words(engine, keys):
    """test/test_api_gpu.cu:140-159: 64 chains of 5 in-place gates on 8 streams, one Synchronize.
    Must run as 5 dependence levels (<= 6 launch sequences), whole program == oracle word for word."""
    api = engine.api
    K = 64
    rng = np.random.default_rng(71)
    bits = rng.integers(0, 2, size=(3, K)).astype(np.uint8)
    (a, ea), (b, eb), (c, ec) = (_ctxts(api, keys, bits[i], 0, 7100 + i) for i in range(3))
    sts = [api.Stream() for _ in range(8)]
    for s in sts:
        s.Create()
    api.sched_stats(reset=True)
    for i in range(K):             # chain by chain: depth-first issue order
        st = sts[i % 8]
        api.Nand(a[i], a[i], b[i], st)
        api.Or(a[i], a[i], b[i], st)
        api.Xor(a[i], a[i], c[i], st)
        api.Not(a[i], a[i], st)
        api.Mux(a[i], a[i], b[i], c[i], st)
    api.Synchronize()
    stats = api.sched_stats()
    assert stats.gates == 5 * K
    assert stats.launch_sequences <= 6, f"{stats.launch_sequences} launch sequences for a 5-level program"
    w = keys.gate_batch(O("NAND"), 0, ea, eb)
    w = keys.gate_batch(O("OR"), 0, w, eb)
    w = keys.gate_batch(O("XOR"), 0, w, ec)
    w = keys.gate_batch(O("NOT"), 0, w)
    w = keys.gate_batch(O("MUX"), 0, w, eb, ec)
    assert np.array_equal(_host(a), w)
    for s in sts:
        s.Destroy()


@pytest.mark.parametrize("rename", [0, 1])
def test_ripple_adders_levels_and_words(engine, keys, rename):
    """16 8-bit ripple-carry adders issued bit by bit, one stream each (640 dependent gates,
    tests/cpp/test_gate_api.cpp RippleAdders): <= 40 launch sequences, sums == oracle words.  With "sched_rename" (the
    default) the re-used temporaries stop ordering the program (t1 is overwritten while its readers are still on record: it
    takes a fresh device buffer) and only the carry chain is left: two levels per bit, plus the one level of Copy gates that
    brings the renamed values back to the ciphertexts' own device buffers (`tlwedevices`, include/cufhe_gpu.cuh:80-84)
    before Synchronize returns -- read back here from those very pointers."""
    api = engine.api
    api.set_option("sched_rename", rename)
    try:
        _ripple_adders(engine, keys, rename)
    finally:
        api.set_option("sched_rename", 1)


def test_ripple_adders_scheduled_gate_by_gate_on_two_lanes(engine, keys):
    """The same adders -- 48 of them, 1920 dependent gates -- with the flush scheduled gate by gate on two lanes ("sched_two_lane" 2:
    whenever eligible; "cus_override" 24 makes the lanes narrow enough for this size: chain steps of 24 rotations on the paired
    low-latency kernel beside bulk chunks of 96 on the batch kernel, on two internal streams): every sum, every carry and every
    temporary word for word the oracle's, tlwedevices included."""
    api = engine.api
    api.set_option("cus_override", 24)
    api.set_option("sched_two_lane", 2)
    try:
        stats = _ripple_adders(engine, keys, 1, A=48, B=8)
        assert stats.two_lane_groups == 1 and stats.two_lane_launches >= 2 * 8, (stats.two_lane_groups, stats.two_lane_launches)
    finally:
        api.set_option("sched_two_lane", 1)
        api.set_option("cus_override", 0)


def _ripple_adders(engine, keys, rename, A=16, B=8):
    api = engine.api
    rng = np.random.default_rng(72)
    va, vb = rng.integers(0, 256, A), rng.integers(0, 256, A)
    xb = np.array([[(va[i] >> k) & 1 for k in range(B)] for i in range(A)], np.uint8).ravel()
    yb = np.array([[(vb[i] >> k) & 1 for k in range(B)] for i in range(A)], np.uint8).ravel()
    x, ex = _ctxts(api, keys, xb, 0, 7201)
    y, ey = _ctxts(api, keys, yb, 0, 7202)
    carry, ecarry = _ctxts(api, keys, np.zeros(A, np.uint8), 0, 7203)
    sums = [api.Ctxt(0) for _ in range(A * B)]
    t1 = [api.Ctxt(0) for _ in range(A)]
    t2 = [api.Ctxt(0) for _ in range(A)]
    sts = [api.Stream() for _ in range(A)]
    for s in sts:
        s.Create()
    api.sched_stats(reset=True)
    for k in range(B):
        for i in range(A):
            X, Y, S, C, st = x[i * B + k], y[i * B + k], sums[i * B + k], carry[i], sts[i]
            api.Xor(t1[i], X, Y, st)
            api.Xor(S, t1[i], C, st)
            api.And(t2[i], t1[i], C, st)
            api.And(t1[i], X, Y, st)          # overwrites t1 after its readers
            api.Or(C, t1[i], t2[i], st)       # in place on the carry
    api.Synchronize()
    stats = api.sched_stats()
    assert stats.gates == 5 * A * B
    assert stats.launch_sequences <= (2 * B + 3 if rename else 40), f"{stats.launch_sequences} launch sequences"
    assert (stats.renames > 0) == bool(rename) and (stats.home_copies > 0) == bool(rename)
    # the pointers published at construction hold the values, renamed on the way or not (every gate here is a copying one:
    # tlwehost has the same words)
    for c in t1 + t2 + carry:
        dev = np.empty(ol.n + 1, np.uint32)
        L = engine._lib
        L.check(L.lib.cufhe_amd_memcpy_d2h(0, None, dev.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(c.tlwedevices[0]), dev.size * 4))
        L.check(L.lib.cufhe_amd_stream_synchronize(0, None))
        assert np.array_equal(dev, c.tlwehost), "tlwedevices[0] does not hold the ciphertext's value after Synchronize"
    # the same program on the oracle, bit by bit (batched over the adders)
    ex, ey = ex.reshape(A, B, -1), ey.reshape(A, B, -1)
    wc = ecarry
    want_sums = np.zeros((A, B, ol.n + 1), np.uint32)
    for k in range(B):
        w1 = keys.gate_batch(O("XOR"), 0, ex[:, k], ey[:, k])
        want_sums[:, k] = keys.gate_batch(O("XOR"), 0, w1, wc)
        w2 = keys.gate_batch(O("AND"), 0, w1, wc)
        w1 = keys.gate_batch(O("AND"), 0, ex[:, k], ey[:, k])
        wc = keys.gate_batch(O("OR"), 0, w1, w2)
    assert np.array_equal(_host(sums).reshape(A, B, -1), want_sums)
    assert np.array_equal(_host(carry), wc)
    got = [sum(int(keys.decrypt(sums[i * B + k].tlwehost, 0)[0]) << k for k in range(B)) +
           (int(keys.decrypt(carry[i].tlwehost, 0)[0]) << B) for i in range(A)]
    assert got == [int(va[i] + vb[i]) for i in range(A)]
    for s in sts:
        s.Destroy()
    return stats


def test_g_gate_flush_copy_poll_then_copying_gate(engine, keys):
    """A g-gate result fetched with CtxtCopyD2H and observed through a StreamQuery poll must be what a
    later copying gate reads (the reference runs D2H then H2D on the stream): words, both ways."""
    api = engine.api
    st = api.Stream()
    st.Create()
    (a, b, e), enc = _ctxts(api, keys, [1, 0, 1], 0, 7301)
    c, d = api.Ctxt(0), api.Ctxt(0)
    api.CtxtCopyH2D(a, st); api.CtxtCopyH2D(b, st)
    api.gNand(c, a, b, st)                 # c (device only) = NAND(a, b)
    api.Flush(0)
    api.CtxtCopyD2H(c, st)                 # its host copy becomes current only through this
    while not api.StreamQuery(st):
        pass
    w_c = keys.gate_batch(O("NAND"), 0, enc[0:1], enc[1:2])
    assert np.array_equal(c.tlwehost, w_c[0])
    api.Xor(d, c, e, st)                   # copying gate: must see NAND(a, b), not a stale tlwehost
    api.Synchronize()
    assert np.array_equal(d.tlwehost, keys.gate_batch(O("XOR"), 0, w_c, enc[2:3])[0])
    # a g-gate reading `a` followed, in the same recorded program, by a copying gate whose upload rewrites
    # a's device buffer from an edited tlwehost: the g-gate must have read the old device value
    new_a = keys.encrypt([0], 0, seed=7302)
    f, g = api.Ctxt(0), api.Ctxt(0)
    api.gOr(f, a, b, st)                   # reads a's device buffer (= enc[0])
    a.tlwehost[:] = new_a[0]
    api.And(g, a, e, st)                   # uploads the new a
    api.CtxtCopyD2H(f, st)
    api.Synchronize()
    assert np.array_equal(f.tlwehost, keys.gate_batch(O("OR"), 0, enc[0:1], enc[1:2])[0])
    assert np.array_equal(g.tlwehost, keys.gate_batch(O("AND"), 0, new_a, enc[2:3])[0])
    st.Destroy()


def test_intensive_polling_shares_inputs(engine, keys):
    """test/test_intensive.cc:21-128 in small: 64 streams x 4 rounds of Nand / Mux on three SHARED
    inputs, refilled by polling StreamQuery; every polled result must be the oracle's words, the
    inputs are uploaded once, and the gates run as a few large launches."""
    api = engine.api
    (in0, in1, inc), enc = _ctxts(api, keys, [1, 1, 0], 0, 7401)
    S, R = 64, 4
    outs = [api.Ctxt(0) for _ in range(S)]
    sts = [api.Stream() for _ in range(S)]
    for s in sts:
        s.Create()
    w_nand = keys.gate_batch(O("NAND"), 0, enc[0:1], enc[1:2])[0]
    w_mux = keys.gate_batch(O("MUX"), 0, enc[2:3], enc[1:2], enc[0:1])[0]
    api.sched_stats(reset=True)
    rnd = [0] * S
    for i in range(S):
        api.Nand(outs[i], in0, in1, sts[i])
    done = 0
    while done < S:
        for i in range(S):
            if rnd[i] >= R or not api.StreamQuery(sts[i]):
                continue
            assert np.array_equal(outs[i].tlwehost, w_nand if rnd[i] % 2 == 0 else w_mux), (i, rnd[i])
            rnd[i] += 1
            if rnd[i] == R:
                done += 1
            elif rnd[i] % 2:
                api.Mux(outs[i], inc, in1, in0, sts[i])
            else:
                api.Nand(outs[i], in0, in1, sts[i])
    api.Synchronize()
    stats = api.sched_stats()
    assert stats.gates == S * R and stats.uploads <= 6 and stats.launch_sequences <= 3 * R, \
        (stats.gates, stats.uploads, stats.launch_sequences)
    for s in sts:
        s.Destroy()


def test_config2_mixed_32768_per_gate_api_256_streams(engine, keys, oracle):
    """BASELINE configs[2] at full size on one GPU through the per-gate API: 32 768 mixed
    AND/OR/XOR/NAND gates round-robin over 256 streams (test/test_util.h:36-62), Synchronize, decrypt
    every output against the truth table (test/test_util.h:75-94) and compare 64 sampled outputs word
    for word with the oracle."""
    api = engine.api
    count, nst = 32768, 256
    rng = np.random.default_rng(43)
    bits = rng.integers(0, 2, size=(2, count)).astype(np.uint8)
    enc = [keys.encrypt(bits[i], 0, seed=4300 + i) for i in range(2)]
    ops = np.array([[O("AND"), O("OR"), O("XOR"), O("NAND")][g % 4] for g in range(count)], np.int32)
    fns = {O("AND"): api.And, O("OR"): api.Or, O("XOR"): api.Xor, O("NAND"): api.Nand}
    sts = [api.Stream() for _ in range(nst)]
    for s in sts:
        s.Create()
    cin = [[api.Ctxt(0) for _ in range(count)] for _ in range(2)]
    for i in range(2):
        for g in range(count):
            cin[i][g].tlwehost[:] = enc[i][g]
    outs = [api.Ctxt(0) for _ in range(count)]
    api.sched_stats(reset=True)
    for g in range(count):
        fns[int(ops[g])](outs[g], cin[0][g], cin[1][g], sts[g % nst])
    api.Synchronize()
    stats = api.sched_stats()
    assert stats.gates == count and stats.launch_sequences <= count // 2048 + 1
    got = _host(outs)
    exp = np.array([ol.truth(oracle, int(ops[g]), bits[0, g], bits[1, g]) for g in range(count)], np.uint8)
    assert np.array_equal(keys.decrypt(got, 0), exp)
    idx = np.arange(5, count, count // 64)[:64]
    want = keys.gate_batch(ops[idx], 0, enc[0][idx], enc[1][idx])
    assert np.array_equal(got[idx], want)
    for s in sts:
        s.Destroy()
    for lst in (cin[0], cin[1], outs):
        for c in lst:
            c.release()


def test_config2_mixed_32768_gate_batch(engine, keys, oracle):
    """The same 32 768 mixed gates through the native batched entry (cufhe_amd_gate_batch)."""
    count = 32768
    rng = np.random.default_rng(43)
    bits = rng.integers(0, 2, size=(2, count)).astype(np.uint8)
    enc = [keys.encrypt(bits[i], 0, seed=4300 + i) for i in range(2)]
    ops = np.array([[O("AND"), O("OR"), O("XOR"), O("NAND")][g % 4] for g in range(count)], np.int32)
    d = [engine.api.DeviceBuffer(e.size).upload(e) for e in enc]
    dout = engine.api.DeviceBuffer(count * (ol.n + 1))
    engine.gate_batch(ops, 0, dout, d[0], d[1], count=count)
    got = dout.download().reshape(count, -1)
    exp = np.array([ol.truth(oracle, int(ops[g]), bits[0, g], bits[1, g]) for g in range(count)], np.uint8)
    assert np.array_equal(keys.decrypt(got, 0), exp)
    idx = np.arange(5, count, count // 64)[:64]
    assert np.array_equal(got[idx], keys.gate_batch(ops[idx], 0, enc[0][idx], enc[1][idx]))


@pytest.mark.parametrize("op", ["MUX", "NMUX"])
def test_config3_mux_4096(engine, keys, oracle, op):
    """BASELINE configs[3] at full size: 4096 MUX (and NMUX) gates = 8192 blind rotations + 4096 key
    switches (src/bootstrap_gpu.cu:515-588); decrypt all, 64 sampled outputs == oracle words."""
    count = 4096
    rng = np.random.default_rng(44)
    bits = rng.integers(0, 2, size=(3, count)).astype(np.uint8)
    enc = [keys.encrypt(bits[i], 0, seed=4400 + i) for i in range(3)]
    d = [engine.api.DeviceBuffer(e.size).upload(e) for e in enc]
    dout = engine.api.DeviceBuffer(count * (ol.n + 1))
    engine.gate_batch(O(op), 0, dout, d[0], d[1], d[2], count=count)
    got = dout.download().reshape(count, -1)
    exp = np.array([ol.truth(oracle, O(op), bits[0, g], bits[1, g], bits[2, g]) for g in range(count)], np.uint8)
    assert np.array_equal(keys.decrypt(got, 0), exp)
    idx = np.arange(3, count, count // 64)[:64]
    assert np.array_equal(got[idx], keys.gate_batch(O(op), 0, enc[0][idx], enc[1][idx], enc[2][idx]))


def test_trlwe_level_ops_are_scheduled(engine, keys, oracle):
    """GateBootstrappingTLWE2TRLWElvl01NTT -> Refresh -> SampleExtractAndKeySwitch chains (test/test_perf.cc:36-87,
    src/cufhe_gates_gpu.cu:86-146) on 8 streams: recorded like gates, three launch sequences for the whole
    program, every intermediate and final value == the oracle's words."""
    api = engine.api
    K = 40
    rng = np.random.default_rng(81)
    bits = rng.integers(0, 2, K).astype(np.uint8)
    ins, enc = _ctxts(api, keys, bits, 0, 8100)
    t = [api.Trlwe() for _ in range(K)]
    r = [api.Trlwe() for _ in range(K)]
    outs = [api.Ctxt(0) for _ in range(K)]
    sts = [api.Stream() for _ in range(8)]
    for s in sts:
        s.Create()
    api.sched_stats(reset=True)
    for i in range(K):
        st = sts[i % 8]
        api.GateBootstrappingTLWE2TRLWElvl01NTT(t[i], ins[i], st)
        api.Refresh(r[i], t[i], st)
        api.SampleExtractAndKeySwitch(outs[i], r[i], st)
    api.Synchronize()
    stats = api.sched_stats()
    assert stats.gates == 3 * K and stats.launch_sequences <= 3, (stats.gates, stats.launch_sequences)
    for i in range(K):
        acc = np.zeros(2 * ol.N, np.uint32)
        oracle.orc_blind_rotate(keys.ek, acc, np.ascontiguousarray(enc[i]), -1)
        assert np.array_equal(t[i].trlwehost, acc), i
        ref = np.zeros(2 * ol.N, np.uint32)
        oracle.orc_refresh(keys.ek, ref, acc)
        assert np.array_equal(r[i].trlwehost, ref), i
        t0 = np.zeros(ol.n + 1, np.uint32)
        oracle.orc_sample_extract_keyswitch(keys.ek, t0, ref)
        assert np.array_equal(outs[i].tlwehost, t0), i
    assert np.array_equal(keys.decrypt(_host(outs), 0), bits)
    for s in sts:
        s.Destroy()


def test_per_gate_api_error_paths(engine, keys):
    """Misuse is reported as a negative status (an exception here, print + exit(-1) in the C++ shim like the reference's
    CuSafeCall, include/details/error_gpu.cuh:40-60), never as a crash or a silently wrong launch."""
    api = engine.api
    st = api.Stream()
    st.Create()
    a0, b0, o0 = api.Ctxt(0), api.Ctxt(0), api.Ctxt(0)
    a1 = api.Ctxt(1)
    t = api.Trlwe()
    with pytest.raises(engine.CufheAmdError):
        api.Nand(o0, a0, a1, st)                               # operands of different levels
    with pytest.raises(engine.CufheAmdError):
        api._gate(99, True, o0, [a0, b0], st)                  # unknown op
    with pytest.raises(engine.CufheAmdError):
        api._trlwe_op(api.TL_REFRESH, True, t, a0, st)         # Refresh takes a TRLWE
    with pytest.raises(engine.CufheAmdError):
        api.Nand(t, a0, b0, st)                                # gates do not write TRLWEs
    o0.release()
    with pytest.raises(Exception):
        api.Nand(o0, a0, b0, st)                               # released handle (None)
    # the scheduler is still healthy afterwards
    a0.tlwehost[:] = keys.encrypt([1], 0, seed=1)[0]
    b0.tlwehost[:] = keys.encrypt([1], 0, seed=2)[0]
    o = api.Ctxt(0)
    api.Nand(o, a0, b0, st)
    api.Synchronize()
    assert keys.decrypt(o.tlwehost, 0)[0] == 0
    st.Destroy()


@pytest.mark.parametrize("seed", range(int(os.environ.get("CUFHE_AMD_STRESS_SEEDS", "1"))))
@pytest.mark.parametrize("rename", [0, 1])
def test_random_program_matches_in_order_oracle(engine, keys, oracle, rename, seed):
    """(rename: the same with "sched_rename", outputs taking fresh device buffers; CUFHE_AMD_STRESS_SEEDS=n runs n programs.)
    A seeded random program through the per-gate API on the real device -- copying gates and g-gates of both levels,
    in-place outputs, shared inputs, explicit copies, Flush, StreamQuery polls -- against an in-order interpreter whose
    gates are the CPU oracle's: every tlwehost must hold the oracle's words at the end (the CPU twin of this test,
    with a stubbed device, is tests/test_sched_model.py)."""
    api = engine.api
    api.set_option("sched_rename", rename)
    rng = np.random.default_rng(2025 + rename + 1000 * seed)
    nct = 14
    cts = {0: [api.Ctxt(0) for _ in range(nct)], 1: [api.Ctxt(1) for _ in range(nct // 2)]}
    host = {}     # model: the eventual tlwehost of every ciphertext
    dev = {}      # model: its device buffer (None = undefined)
    for lvl, lst in cts.items():
        enc = keys.encrypt(rng.integers(0, 2, len(lst)).astype(np.uint8), lvl, seed=9000 + lvl)
        for c, e in zip(lst, enc):
            c.tlwehost[:] = e
            host[id(c)] = e.copy()
            dev[id(c)] = None
    sts = [api.Stream() for _ in range(5)]
    for s in sts:
        s.Create()
    two = ["NAND", "NOR", "XNOR", "AND", "OR", "XOR", "ANDNY", "ANDYN", "ORNY", "ORYN"]
    fn = {n: (getattr(api, n[0] + n[1:].lower().replace("ny", "NY").replace("yn", "YN")),
              getattr(api, "g" + n[0] + n[1:].lower().replace("ny", "NY").replace("yn", "YN"))) for n in two}

    def run_oracle(op, lvl, ins):
        arrs = [np.ascontiguousarray(x).reshape(1, -1) for x in ins] + [None] * (3 - len(ins))
        return keys.gate_batch(op, lvl, arrs[0], arrs[1], arrs[2], threads=1)[0]

    for step in range(110):
        lvl = 1 if rng.random() < 0.25 else 0
        pool = cts[lvl]
        st = sts[rng.integers(len(sts))]
        r = rng.random()
        if r < 0.55:                                              # copying gate
            out = pool[rng.integers(len(pool))]
            a = out if rng.random() < 0.2 else pool[rng.integers(len(pool))]
            b, c = pool[rng.integers(len(pool))], pool[rng.integers(len(pool))]
            kind = rng.random()
            for x in (a, b, c):
                pass
            if kind < 0.12:
                dev[id(a)] = host[id(a)].copy()
                res = run_oracle(ol.OPS.index("NOT"), lvl, [dev[id(a)]])
                api.Not(out, a, st)
            elif kind < 0.3:
                for x in (a, b, c):
                    dev[id(x)] = host[id(x)].copy()
                res = run_oracle(ol.OPS.index("MUX"), lvl, [dev[id(a)], dev[id(b)], dev[id(c)]])
                api.Mux(out, a, b, c, st)
            else:
                name = two[rng.integers(len(two))]
                for x in (a, b):
                    dev[id(x)] = host[id(x)].copy()
                res = run_oracle(ol.OPS.index(name), lvl, [dev[id(a)], dev[id(b)]])
                fn[name][0](out, a, b, st)
            dev[id(out)] = res
            host[id(out)] = res.copy()
        elif r < 0.75:                                            # g-gate on defined device buffers
            defined = [x for x in pool if dev[id(x)] is not None]
            if len(defined) < 2:
                continue
            a, b = defined[rng.integers(len(defined))], defined[rng.integers(len(defined))]
            out = a if rng.random() < 0.25 else pool[rng.integers(len(pool))]
            name = two[rng.integers(len(two))]
            res = run_oracle(ol.OPS.index(name), lvl, [dev[id(a)], dev[id(b)]])
            fn[name][1](out, a, b, st)
            dev[id(out)] = res
        elif r < 0.82:
            x = pool[rng.integers(len(pool))]
            api.CtxtCopyH2D(x, st)
            dev[id(x)] = host[id(x)].copy()
        elif r < 0.9:
            defined = [x for x in pool if dev[id(x)] is not None]
            if defined:
                x = defined[rng.integers(len(defined))]
                api.CtxtCopyD2H(x, st)
                host[id(x)] = dev[id(x)].copy()
        elif r < 0.96:
            api.StreamQuery(st)
        else:
            api.Flush(0)
    api.Synchronize()
    api.set_option("sched_rename", 1)
    for lvl, lst in cts.items():
        for i, c in enumerate(lst):
            assert np.array_equal(c.tlwehost, host[id(c)]), f"level {lvl} ciphertext {i}"
    for s in sts:
        s.Destroy()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("CUFHE_AMD_STRESS_SEEDS", "2")))))
def test_random_netlist_on_two_lanes_matches_in_order_oracle(engine, keys, seed):
    """A seeded random NETLIST on the real device with the flush scheduled gate by gate on two lanes ("sched_two_lane" 2, "cus_override"
    16: chain steps of 16 rotations beside bulk chunks of 64 on two internal streams): inputs uploaded once, then 260 device-resident gates
    whose operands are drawn from everything computed so far -- long chains beside wide independent work, temporaries re-used (renamed),
    in-place gates, Mux (two rotations), Not -- every value fetched at the end.  Every tlwehost must hold the words of the in-order
    interpreter whose gates are the CPU oracle's (the CPU twin with a stubbed device: tests/host/sched_harness.cpp dag_program)."""
    api = engine.api
    api.set_option("cus_override", 16)
    api.set_option("sched_two_lane", 2)
    try:
        rng = np.random.default_rng(4100 + seed)
        n_in, n_tmp, n_gates = 10, 14, 260
        cts = [api.Ctxt(0) for _ in range(n_in + n_tmp)]
        enc = keys.encrypt(rng.integers(0, 2, n_in).astype(np.uint8), 0, seed=9100 + seed)
        sts = [api.Stream() for _ in range(4)]
        for s in sts:
            s.Create()
        dev = [None] * len(cts)
        for i in range(n_in):
            cts[i].tlwehost[:] = enc[i]
            api.CtxtCopyH2D(cts[i], sts[i % 4])
            dev[i] = enc[i].copy()
        two = ["NAND", "NOR", "XNOR", "AND", "OR", "XOR", "ANDNY", "ANDYN", "ORNY", "ORYN"]
        gfn = {n: getattr(api, "g" + n[0] + n[1:].lower().replace("ny", "NY").replace("yn", "YN")) for n in two}

        def oracle(op, ins):
            arrs = [np.ascontiguousarray(x).reshape(1, -1) for x in ins] + [None] * (3 - len(ins))
            return keys.gate_batch(op, 0, arrs[0], arrs[1], arrs[2], threads=1)[0]
        api.Synchronize()
        api.sched_stats(reset=True)
        last = 0
        for k in range(n_gates):
            defined = [i for i in range(len(cts)) if dev[i] is not None]
            a = last if rng.random() < 0.65 else defined[rng.integers(len(defined))]       # most gates extend what was just computed
            b, c = defined[rng.integers(len(defined))], defined[rng.integers(len(defined))]
            o = a if rng.random() < 0.15 else n_in + rng.integers(n_tmp)
            st = sts[rng.integers(4)]
            r = rng.random()
            if r < 0.05:
                res = oracle(ol.OPS.index("NOT"), [dev[a]])
                api.gNot(cts[o], cts[a], st)
            elif r < 0.15:
                res = oracle(ol.OPS.index("MUX"), [dev[a], dev[b], dev[c]])
                api.gMux(cts[o], cts[a], cts[b], cts[c], st)
            else:
                name = two[rng.integers(len(two))]
                res = oracle(ol.OPS.index(name), [dev[a], dev[b]])
                gfn[name](cts[o], cts[a], cts[b], st)
            dev[o] = res
            last = o
        for i in range(len(cts)):
            if dev[i] is not None:
                api.CtxtCopyD2H(cts[i], sts[i % 4])
        api.Synchronize()
        stats = api.sched_stats()
        assert stats.two_lane_groups >= 1 and stats.two_lane_launches >= 8, (stats.two_lane_groups, stats.two_lane_launches, stats.levels)
        for i in range(len(cts)):
            if dev[i] is not None:
                assert np.array_equal(cts[i].tlwehost, dev[i]), f"ciphertext {i} differs from the in-order oracle"
        for s in sts:
            s.Destroy()
    finally:
        api.set_option("sched_two_lane", 1)
        api.set_option("cus_override", 0)
