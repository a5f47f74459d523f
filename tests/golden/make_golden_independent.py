#!/usr/bin/env python3
"""make_golden_independent.py -- a SECOND, independent generator of full-size gate fixtures.

Every other fixture under tests/golden/ is written by oracle/ (the C restatement the GPU words are compared with), so a
misreading of the reference shared by the oracle and the kernels would pass unnoticed.  This script shares no code with
oracle/ or cufhe_amd/: numpy and Python integers only, the external product as an exact SCHOOLBOOK negacyclic
convolution in 64-bit integers (no transform, no prime, no floating point), written from the reference's text:

  gate constants, pre-add            src/bootstrap_gpu.cu:402-421 (__HomGate__), :424-512 (the ten gates), :515-588 (Mux, NMux)
  modulus switch, test vector         include/gatebootstrapping_gpu.cuh:10-52, :287-345 (__BlindRotatePreAdd__)
  (X^abar - 1) acc, gadget digits     include/gatebootstrapping_gpu.cuh:140-181
  key indexing [step][row][out][N]    include/gatebootstrapping_gpu.cuh:207-217, src/bootstrap_gpu.cu:43-49
  sample extract at index 0           src/bootstrap_gpu.cu:366-381
  key switch                          include/keyswitch_gpu.cuh:13-23 (iksoffsetgen), :83-134 (KeySwitchFromTLWE)

It writes tests/golden/golden_independent_v5.json: on the BASELINE set (n = 630, N = 1024) all ten two-input gates, MUX and
NMUX on level-0 ciphertexts (blind rotate, then key switch), NAND on level-1 ciphertexts (the other order:
IdentityKeySwitchPreAdd, then __BlindRotate__; src/bootstrap_gpu.cu:383-400, include/keyswitch_gpu.cuh:136-188), and one
NAND through the N = 2048 / 64-bit ring (the reference's templates instantiated at lvl02 / lvl20, as DESIGN.md 5a
defines that path); NAND, XOR and MUX on the k = 2 / N = 512 set and NAND, ORYN on the n = 500 / l = 2 / Bg = 2^10 set
(DESIGN.md 5b).  Since v4 also the inputs at which a misreading is most likely to hide (the cases the oracle tests feed the
kernels, tests/test_gpu_parity.py): runs of abar = 0 (include/gatebootstrapping_gpu.cuh:157-181: (X^0 - 1) acc = 0, the digits of the
bare offset), bbar = 2N / N / 1 (:29-52, the three branches of the test vector), input words 0x7FFFFFFF, a key whose first two CMux
steps are all 0x80000000 and whose other words are drawn from {0x80000000, 0x7FFFFFFF, 0, 0xFFFFFFFF, 0x80000001} (the external
product's sums closest to the exactness bound), MUX on level-1 ciphertexts (src/bootstrap_gpu.cu:706-743: two key switches, two
blind rotations, the sum of the two EXTRACTED ciphertexts) and Not / Copy (:681-703).  Since v5 the TRLWE-level primitives of
src/cufhe_gates_gpu.cu:69-146 on the BASELINE set: CMUXNTT (src/bootstrap_gpu.cu:160-285: decomposition of c1 - c0, external product
with a TRGSW of arbitrary words, + c0), the bootstrap to a TRLWE without sample extract (__BlindRotateGlobal__, :317-323), sample
extract + key switch (__SEIandKS__, src/keyswitch_gpu.cu:26-40) and Refresh (:325-364: sample extract, key switch, blind rotation
with the test vector taken from the key-switched ciphertext -- the reference reads an uninitialised buffer there; the
oracle and the kernels define it this way).  Keys are uniform random words from a seeded numpy generator (the path is data-independent: any
key words define a word-level check); the fixture stores the seeds, a sha256 of each generated key, the inputs and the
expected output words.  Takes a few minutes; run in the build container only:  python tests/golden/make_golden_independent.py
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

MU0 = 1 << 29            # lvl0 / lvl1 message scale


class Ring:
    """the target ring of a blind rotation and the key switch back to lvl0"""

    def __init__(self, N, nbit, l, Bgbit, bits, mu, t, basebit, n=630, k=1, small_modulus=False):
        self.N, self.nbit, self.l, self.Bgbit, self.bits, self.mu, self.t, self.basebit = N, nbit, l, Bgbit, bits, mu, t, basebit
        self.n, self.k = n, k                            # lvl0 dimension; mask polynomials of the ring
        self.small_modulus = small_modulus               # the reference built with -DUSE_SMALL_NTT_MODULUS (below)
        self.mask = (1 << bits) - 1
        i = np.arange(N)[:, None]
        m = np.arange(N)[None, :]
        self.conv_idx = (i - m) % N                      # (d * k)[i] = sum_m d[(i - m) mod N] k[m] * (-1 if i < m)
        self.conv_sign = np.where(i >= m, 1, -1).astype(np.int64)


LVL1 = Ring(N=1024, nbit=10, l=3, Bgbit=6, bits=32, mu=1 << 29, t=8, basebit=2)
LVL2 = Ring(N=2048, nbit=11, l=4, Bgbit=9, bits=64, mu=1 << 61, t=7, basebit=2)
# the other parameter sets compiled into the library (cufhe_amd/csrc/kernels_ps.hip.h; the reference selects them at build
# time, CMakeLists.txt:8-24; k > 1: src/bootstrap_gpu.cu:402-421, include/gatebootstrapping_gpu.cuh:153-224)
K2N512 = Ring(N=512, nbit=9, l=3, Bgbit=6, bits=32, mu=1 << 29, t=8, basebit=2, n=630, k=2)
CGGI16 = Ring(N=1024, nbit=10, l=2, Bgbit=10, bits=32, mu=1 << 29, t=8, basebit=2, n=500, k=1)
# The BASELINE numbers as the reference computes them when built with -DUSE_SMALL_NTT_MODULUS (CMakeLists.txt:12,26-28): the
# external product is taken modulo SMALL_P on a key switched to the P discretisation of the torus, and each CMux increment is
# switched back (include/ntt_gpu/ntt_small_modulus.cuh:36-68,147-177; src/bootstrap_gpu.cu:50-66;
# include/gatebootstrapping_gpu.cuh:133-137,185-193,236-248).  Arithmetic mod P is arithmetic mod P whatever computes it:
# here the schoolbook sum in exact integers, reduced with Python's %.
SMALLMOD = Ring(N=1024, nbit=10, l=3, Bgbit=6, bits=32, mu=1 << 29, t=8, basebit=2, small_modulus=True)
SMALL_P = 625 * 2**20 + 1                        # small_ntt::P
SMALL_INV_MODSWITCH_MUL = 2**63 // SMALL_P       # small_ntt::INV_MODSWITCH_MUL


def torus32_to_ntt_mod(a):
    """round(a P / 2^32) = (a P + 2^31) >> 32 on an array of torus words (int64 holds a P < 2^62)"""
    return (a.astype(np.int64) * SMALL_P + (1 << 31)) >> 32


def ntt_mod_to_torus32(r):
    """round(r 2^32 / P) for the residue r in [0, P): (r * (2^63 / P) + 2^30) >> 31, truncated to 32 bits (python integers)"""
    return ((r * SMALL_INV_MODSWITCH_MUL + (1 << 30)) >> 31) & 0xFFFFFFFF


def modswitch(phase32, R):
    """modSwitchFromTorus: the top nbit + 1 bits of a 32-bit phase"""
    return (phase32 & 0xFFFFFFFF) >> (32 - 1 - R.nbit)


def rotated_test_vector(bar, R):
    """RotatedTestVector: the k mask polynomials 0, b = X^bar * (mu + mu X + ...), python ints mod 2^bits"""
    b = []
    for i in range(R.N):
        if bar == 2 * R.N:
            b.append(R.mu)
        else:
            neg = (i < (bar & (R.N - 1))) ^ ((bar >> R.nbit) & 1)
            b.append((-R.mu) & R.mask if neg else R.mu)
    return [[0] * R.N for _ in range(R.k)] + [b]


def gadget_digits(acc_j, abar, R):
    """digits of (X^abar - 1) acc_j: list over d of int64 arrays of N signed digits"""
    N = R.N
    offset = sum((1 << (R.Bgbit - 1)) << (R.bits - i * R.Bgbit) for i in range(1, R.l + 1))
    roundoffset = 1 << (R.bits - R.l * R.Bgbit - 1)
    digs = [np.zeros(N, np.int64) for _ in range(R.l)]
    for i in range(N):
        temp = acc_j[(i - abar) & (N - 1)]
        if (i < (abar & (N - 1))) ^ ((abar >> R.nbit) & 1):
            temp = -temp
        temp = (temp - acc_j[i] + offset + roundoffset) & R.mask
        for d in range(R.l):
            digs[d][i] = ((temp >> (R.bits - (d + 1) * R.Bgbit)) & ((1 << R.Bgbit) - 1)) - (1 << (R.Bgbit - 1))
    return digs


def blind_rotate(c, bk, R):
    """__BlindRotatePreAdd__ after the pre-add: c = n + 1 words of the linear combination (32-bit python ints).
    bk: uint32 or uint64 array [n][(k+1) l][k+1][N].  Returns the accumulator [k+1][N] (python ints)."""
    N, K1 = R.N, R.k + 1
    acc = rotated_test_vector(2 * N - modswitch(c[R.n], R), R)
    roundoffset = 1 << (32 - 2 - R.nbit)
    rows = K1 * R.l
    for i in range(R.n):
        abar = modswitch(c[i] + roundoffset, R)
        # negacyclic matrices of the (k+1) l digit polynomials, side by side: [N][rows * N]
        T = np.empty((N, rows * N), np.int64)
        for j in range(K1):
            for d, dig in enumerate(gadget_digits(acc[j], abar, R)):
                r = j * R.l + d
                T[:, r * N:(r + 1) * N] = dig[R.conv_idx] * R.conv_sign
        key = bk[i]                                          # [rows][k+1][N]
        for o in range(K1):
            if R.small_modulus:
                col = torus32_to_ntt_mod(key[:, o, :].reshape(rows * N))  # in [0, P]: |sum| < 6144 * 32 * 2^29.3 < 2^47
                s = T @ col
                add = [ntt_mod_to_torus32(int(v) % SMALL_P) for v in s]   # the sum mod P (python's % : in [0, P)), switched back
            elif R.bits == 32:
                col = key[:, o, :].reshape(rows * N).astype(np.int64)     # < 2^32: exact in int64, |sum| < 2^53
                s = T @ col
                add = [int(v) & R.mask for v in s]
            else:
                kv = key[:, o, :].reshape(rows * N)
                lo = (kv & np.uint64(0xFFFFFFFF)).astype(np.int64)
                hi = (kv >> np.uint64(32)).astype(np.int64)                # two 32-bit halves: each |sum| < 2^54
                slo, shi = T @ lo, T @ hi
                add = [(int(a) + (int(b) << 32)) & R.mask for a, b in zip(slo, shi)]
            acc[o] = [(x + y) & R.mask for x, y in zip(acc[o], add)]
        if i % 90 == 0:
            print(f"    step {i}/{R.n}", flush=True)
    return acc


def sample_extract0(acc, R):
    """__SampleExtractIndex__<P, 0>: k N + 1 words"""
    N = R.N
    out = []
    for kk in range(R.k):
        out += [acc[kk][0]] + [(-acc[kk][N - m]) & R.mask for m in range(1, N)]
    return out + [acc[R.k][0]]


def keyswitch(tlwe, ksk, R):
    """KeySwitchFromTLWE: tlwe k N + 1 words of R.bits bits -> n + 1 32-bit words.  ksk: uint32 [k N][t][2][n + 1]"""
    roundoffset = 1 << (R.bits - (1 + R.basebit * R.t))
    decompoffset = sum((1 << (R.basebit - 1)) << (R.bits - i * R.basebit) for i in range(1, R.t + 1))
    kN = R.k * R.N
    res = np.zeros(R.n + 1, np.int64)
    b = tlwe[kN]
    res[R.n] = b if R.bits == 32 else ((b + (1 << 31)) & R.mask) >> 32
    for j in range(kN):
        tmp = (tlwe[j] + decompoffset + roundoffset) & R.mask
        for k in range(R.t):
            val = ((tmp >> (R.bits - (k + 1) * R.basebit)) & ((1 << R.basebit) - 1)) - (1 << (R.basebit - 1))
            if val > 0:
                res -= ksk[j, k, val - 1].astype(np.int64)
            elif val < 0:
                res += ksk[j, k, -val - 1].astype(np.int64)
        res &= 0xFFFFFFFF
    return [int(v) for v in res]


def lincomb(ca, in0, cb, in1, off):
    c = [(ca * int(a) + cb * int(b)) & 0xFFFFFFFF for a, b in zip(in0, in1)]
    c[-1] = (c[-1] + off) & 0xFFFFFFFF
    return c


# (casign, cbsign, offset in units of mu0), src/bootstrap_gpu.cu:424-512
GATES = {"NAND": (-1, -1, 1), "NOR": (-1, -1, -1), "XNOR": (-2, -2, -2), "AND": (1, 1, -1), "OR": (1, 1, 1), "XOR": (2, 2, 2),
         "ANDNY": (-1, 1, -1), "ANDYN": (1, -1, -1), "ORNY": (-1, 1, 1), "ORYN": (1, -1, 1)}


def gate2(op, in0, in1, bk, ksk, R):
    """__HomGate__<brP, mu, iksP, casign, cbsign, offset> on level-0 ciphertexts: blind rotate, extract, key switch"""
    ca, cb, off = GATES[op]
    acc = blind_rotate(lincomb(ca, in0, cb, in1, off * MU0), bk, R)
    return keyswitch(sample_extract0(acc, R), ksk, R)


def gate_mux(inc, in1, in0, bk, ksk, R, negate=False):
    """__MuxBootstrap__: BR(inc + in1 - mu0) + BR(-inc + in0 - mu0) + (0, mu), sample extract, key switch;
    __NMuxBootstrap__: -BR(..) - BR(..) - (0, mu)"""
    a1 = blind_rotate(lincomb(1, inc, 1, in1, -MU0), bk, R)
    a0 = blind_rotate(lincomb(-1, inc, 1, in0, -MU0), bk, R)
    sgn = -1 if negate else 1
    acc = [[(sgn * (x + y)) & R.mask for x, y in zip(a1[j], a0[j])] for j in range(R.k + 1)]
    acc[R.k][0] = (acc[R.k][0] + sgn * R.mu) & R.mask
    return keyswitch(sample_extract0(acc, R), ksk, R)


def gate2_level1(op, in0, in1, bk, ksk, R):
    """__HomGate__<iksP, brP, mu, casign, cbsign, offset> on level-1 ciphertexts (N + 1 words): IdentityKeySwitchPreAdd
    (the key switch of the linear combination), then __BlindRotate__ with the test vector mu, sample extract"""
    ca, cb, off = GATES[op]
    pre = [(ca * int(a) + cb * int(b)) & R.mask for a, b in zip(in0, in1)]
    pre[R.k * R.N] = (pre[R.k * R.N] + off * MU0) & R.mask
    acc = blind_rotate(keyswitch(pre, ksk, R), bk, R)
    return sample_extract0(acc, R)


def random_words(rng, count, bits=32):
    if bits == 32:
        return rng.integers(0, 2**32, size=count, dtype=np.uint64).astype(np.uint32)
    return rng.integers(0, 2**64, size=count, dtype=np.uint64)


def key_for(seed, R):
    rng = np.random.default_rng(seed)
    K1 = R.k + 1
    bk = random_words(rng, R.n * K1 * R.l * K1 * R.N, R.bits).reshape(R.n, K1 * R.l, K1, R.N)
    ksk = random_words(rng, R.k * R.N * R.t * 2 * (R.n + 1)).reshape(R.k * R.N, R.t, 2, R.n + 1)
    return bk, ksk, {"seed": seed, "bk_sha256": hashlib.sha256(bk.tobytes()).hexdigest(), "ksk_sha256": hashlib.sha256(ksk.tobytes()).hexdigest()}


def digits_of(vals, R):
    """gadget digits of a polynomial of torus words (TRLWESubAndDecomposition, src/bootstrap_gpu.cu:162-195): list over d of int64 arrays"""
    offset = sum((1 << (R.Bgbit - 1)) << (R.bits - i * R.Bgbit) for i in range(1, R.l + 1))
    roundoffset = 1 << (R.bits - R.l * R.Bgbit - 1)
    digs = [np.zeros(R.N, np.int64) for _ in range(R.l)]
    for i in range(R.N):
        temp = (int(vals[i]) + offset + roundoffset) & R.mask
        for d in range(R.l):
            digs[d][i] = ((temp >> (R.bits - (d + 1) * R.Bgbit)) & ((1 << R.Bgbit) - 1)) - (1 << (R.Bgbit - 1))
    return digs


def cmux(trgsw, c1, c0, R):
    """__CMUXNTT__ (src/bootstrap_gpu.cu:197-285): res = c0 + trgsw [x] (c1 - c0).  trgsw: uint32 [(k+1) l][k+1][N] torus words;
    c1, c0: TRLWEs as [(k+1)][N] python ints"""
    N, K1 = R.N, R.k + 1
    rows = K1 * R.l
    T = np.empty((N, rows * N), np.int64)
    for j in range(K1):
        diff = [(int(a) - int(b)) & R.mask for a, b in zip(c1[j], c0[j])]
        for d, dig in enumerate(digits_of(diff, R)):
            r = j * R.l + d
            T[:, r * N:(r + 1) * N] = dig[R.conv_idx] * R.conv_sign
    out = []
    for o in range(K1):
        s = T @ trgsw[:, o, :].reshape(rows * N).astype(np.int64)
        out.append([(int(x) + int(v)) & R.mask for x, v in zip(c0[o], s)])
    return out


def refresh(trlwe, bk, ksk, R):
    """Refresh (src/cufhe_gates_gpu.cu:106-124, __SEIandBootstrap2TRLWE__ src/bootstrap_gpu.cu:325-364): sample extract at 0, key
    switch to lvl0, blind rotation with the test vector mu rotated by the KEY-SWITCHED ciphertext's b, no sample extract"""
    return blind_rotate(keyswitch(sample_extract0(trlwe, R), ksk, R), bk, R)


def gate_mux_level1(inc, in1, in0, bk, ksk, R):
    """__MuxBootstrap__<iksP, brP, mu> on level-1 ciphertexts (src/bootstrap_gpu.cu:706-743): key switch of inc + in1 - mu, blind
    rotation, sample extract; the same for -inc + in0 - mu; the two extracted ciphertexts added, mu added to b"""
    def leg(ca, x):
        pre = [(ca * int(a) + int(b)) & R.mask for a, b in zip(inc, x)]
        pre[R.k * R.N] = (pre[R.k * R.N] - MU0) & R.mask
        return sample_extract0(blind_rotate(keyswitch(pre, ksk, R), bk, R), R)
    t1, t0 = leg(1, in1), leg(-1, in0)
    out = [(a + b) & R.mask for a, b in zip(t1, t0)]
    out[R.k * R.N] = (out[R.k * R.N] + R.mu) & R.mask
    return out


EXTREME_WORDS = (0x80000000, 0x7FFFFFFF, 0, 0xFFFFFFFF, 0x80000001)


def key_extreme(seed, R):
    """key words of maximal magnitude: the first two CMux steps all 0x80000000 (-2^31 in a signed reading), the rest drawn from
    EXTREME_WORDS; the key-switching key uniform (drawn after the bootstrapping key, from the same generator)"""
    rng = np.random.default_rng(seed)
    K1 = R.k + 1
    ext = np.array(EXTREME_WORDS, np.uint32)
    bk = ext[rng.integers(0, ext.size, R.n * K1 * R.l * K1 * R.N)]
    bk[: 2 * K1 * R.l * K1 * R.N] = 0x80000000
    bk = bk.reshape(R.n, K1 * R.l, K1, R.N)
    ksk = random_words(rng, R.k * R.N * R.t * 2 * (R.n + 1)).reshape(R.k * R.N, R.t, 2, R.n + 1)
    return bk, ksk, {"seed": seed, "kind": "extreme", "bk_sha256": hashlib.sha256(bk.tobytes()).hexdigest(),
                     "ksk_sha256": hashlib.sha256(ksk.tobytes()).hexdigest()}


def edge_inputs(irng):
    """level-0 operand pairs whose linear combinations hit the corners of modswitch / RotatedTestVector:
       a: NAND (c = -in0 - in1 + mu):  c.a[0..7] = c.a[100] = c.a[629] = 0 -> abar = 0 ;  c.b = 0           -> bbar = 2N
       b: AND  (c =  in0 + in1 - mu):  c.a[0..7] = 0x7FFFFFFF              -> abar = N ;  c.b = 0x80000000 -> bbar = N
       c: OR   (c =  in0 + in1 + mu):  c.a[0..3] = 0xFFFFFFFF (rounds up to abar = 2N, i.e. 0)  ;  c.b = 0xFFFFFFFF -> bbar = 1"""
    prs = []
    for _ in range(3):
        prs.append([random_words(irng, 631).astype(np.int64), random_words(irng, 631).astype(np.int64)])
    a0, a1 = prs[0]
    for i in list(range(8)) + [100, 629]:
        a1[i] = (-a0[i]) & 0xFFFFFFFF                    # -in0 - in1 = 0
    a1[630] = (MU0 - a0[630]) & 0xFFFFFFFF               # -b0 - b1 + mu = 0
    b0, b1 = prs[1]
    for i in range(8):
        b1[i] = (0x7FFFFFFF - b0[i]) & 0xFFFFFFFF
    b1[630] = (0x80000000 + MU0 - b0[630]) & 0xFFFFFFFF
    c0, c1 = prs[2]
    for i in range(4):
        c1[i] = (0xFFFFFFFF - c0[i]) & 0xFFFFFFFF
    c1[630] = (0xFFFFFFFF - MU0 - c0[630]) & 0xFFFFFFFF
    return [[x.astype(np.uint32) for x in p] for p in prs]


def main():
    out = {"format": 5, "generator": "tests/golden/make_golden_independent.py (schoolbook, no code shared with oracle/)", "cases": []}
    t0 = time.time()
    irng = np.random.default_rng(777)
    ins0 = [random_words(irng, 631) for _ in range(3)]
    ins1 = [random_words(irng, 1025) for _ in range(2)]
    ins500 = [random_words(irng, 501) for _ in range(2)]
    # drawn after everything the v3 fixture drew, so that its cases keep their words
    ins1.append(random_words(irng, 1025))
    edges = edge_inputs(irng)
    out["inputs"] = {"level0": [x.tolist() for x in ins0], "level1": [x.tolist() for x in ins1], "level0_n500": [x.tolist() for x in ins500],
                     "level0_edge_a": [x.tolist() for x in edges[0]], "level0_edge_b": [x.tolist() for x in edges[1]],
                     "level0_edge_c": [x.tolist() for x in edges[2]]}
    # the corners are where they are meant to be (python integers, the reference's formulas)
    ca = lincomb(-1, edges[0][0], -1, edges[0][1], MU0)
    assert all(modswitch(ca[i] + (1 << 20), LVL1) == 0 for i in list(range(8)) + [100, 629]) and 2 * 1024 - modswitch(ca[630], LVL1) == 2048
    cb = lincomb(1, edges[1][0], 1, edges[1][1], -MU0)
    assert all(modswitch(cb[i] + (1 << 20), LVL1) == 1024 for i in range(8)) and 2 * 1024 - modswitch(cb[630], LVL1) == 1024
    cc = lincomb(1, edges[2][0], 1, edges[2][1], MU0)
    assert all(modswitch(cc[i] + (1 << 20), LVL1) == 0 for i in range(4)) and 2 * 1024 - modswitch(cc[630], LVL1) == 1
    # --- BASELINE set: n = 630, N = 1024
    bk, ksk, key1 = key_for(20261004, LVL1)
    for op in GATES:
        print(op, "default set, level 0", flush=True)
        out["cases"].append({"set": "default", "level": 0, "op": op, "key": key1, "inputs": "level0", "operands": [0, 1],
                             "expected": gate2(op, ins0[0], ins0[1], bk, ksk, LVL1)})
    for op, neg in (("MUX", False), ("NMUX", True)):
        print(op, "default set, level 0", flush=True)
        out["cases"].append({"set": "default", "level": 0, "op": op, "key": key1, "inputs": "level0", "operands": [0, 1, 2],
                             "expected": gate_mux(ins0[0], ins0[1], ins0[2], bk, ksk, LVL1, negate=neg)})
    print("NAND, default set, level 1", flush=True)
    out["cases"].append({"set": "default", "level": 1, "op": "NAND", "key": key1, "inputs": "level1", "operands": [0, 1],
                         "expected": gate2_level1("NAND", ins1[0], ins1[1], bk, ksk, LVL1)})
    for op, tag in (("NAND", "a"), ("AND", "b"), ("OR", "c")):
        print(op, "default set, level 0, edge inputs", tag, flush=True)
        e = edges["abc".index(tag)]
        out["cases"].append({"set": "default", "level": 0, "op": op, "key": key1, "inputs": "level0_edge_" + tag, "operands": [0, 1],
                             "expected": gate2(op, e[0], e[1], bk, ksk, LVL1)})
    print("MUX, default set, level 1", flush=True)
    out["cases"].append({"set": "default", "level": 1, "op": "MUX", "key": key1, "inputs": "level1", "operands": [0, 1, 2],
                         "expected": gate_mux_level1(ins1[0], ins1[1], ins1[2], bk, ksk, LVL1)})
    # Not / Copy: no bootstrapping (src/bootstrap_gpu.cu:681-703)
    out["cases"].append({"set": "default", "level": 0, "op": "NOT", "key": key1, "inputs": "level0", "operands": [2],
                         "expected": [(-int(v)) & 0xFFFFFFFF for v in ins0[2]]})
    out["cases"].append({"set": "default", "level": 1, "op": "COPY", "key": key1, "inputs": "level1", "operands": [1],
                         "expected": [int(v) for v in ins1[1]]})
    # --- TRLWE-level primitives (BASELINE set): operands are TRLWEs [(k+1) N] / a TRGSW [(k+1) l][k+1][N] of uniform words
    trl = [random_words(irng, 2 * 1024) for _ in range(2)]
    trgsw = random_words(irng, 6 * 2 * 1024)
    out["inputs"]["trlwe"] = [x.tolist() for x in trl]
    out["inputs"]["trgsw"] = [trgsw.tolist()]
    as_trlwe = lambda x: [[int(v) for v in x[:1024]], [int(v) for v in x[1024:]]]
    flat = lambda t: [int(v) for comp in t for v in comp]
    print("CMUXNTT", flush=True)
    out["cases"].append({"set": "default", "level": 2, "op": "CMUXNTT", "key": key1, "inputs": "trlwe", "operands": [0, 1],
                         "expected": flat(cmux(trgsw.reshape(6, 2, 1024), as_trlwe(trl[0]), as_trlwe(trl[1]), LVL1))})
    print("BOOT2TRLWE (lvl0 TLWE -> TRLWE)", flush=True)
    out["cases"].append({"set": "default", "level": 2, "op": "BOOT2TRLWE", "key": key1, "inputs": "level0", "operands": [2],
                         "expected": flat(blind_rotate([int(v) for v in ins0[2]], bk, LVL1))})
    print("SEIKS (TRLWE -> lvl0 TLWE)", flush=True)
    out["cases"].append({"set": "default", "level": 2, "op": "SEIKS", "key": key1, "inputs": "trlwe", "operands": [0],
                         "expected": keyswitch(sample_extract0(as_trlwe(trl[0]), LVL1), ksk, LVL1)})
    print("REFRESH (TRLWE -> TRLWE)", flush=True)
    out["cases"].append({"set": "default", "level": 2, "op": "REFRESH", "key": key1, "inputs": "trlwe", "operands": [1],
                         "expected": flat(refresh(as_trlwe(trl[1]), bk, ksk, LVL1))})
    bkx, kskx, keyx = key_extreme(20261008, LVL1)
    print("NAND, default set, level 0, extreme key words", flush=True)
    out["cases"].append({"set": "default", "level": 0, "op": "NAND", "key": keyx, "inputs": "level0", "operands": [0, 1],
                         "expected": gate2("NAND", ins0[0], ins0[1], bkx, kskx, LVL1)})
    # --- the other compiled parameter sets
    bk, ksk, key = key_for(20261006, K2N512)
    for op in ("NAND", "XOR"):
        print(op, "k2n512 (N = 512, k = 2)", flush=True)
        out["cases"].append({"set": "k2n512", "level": 0, "op": op, "key": key, "inputs": "level0", "operands": [0, 1],
                             "expected": gate2(op, ins0[0], ins0[1], bk, ksk, K2N512)})
    print("MUX, k2n512", flush=True)
    out["cases"].append({"set": "k2n512", "level": 0, "op": "MUX", "key": key, "inputs": "level0", "operands": [0, 1, 2],
                         "expected": gate_mux(ins0[0], ins0[1], ins0[2], bk, ksk, K2N512)})
    print("NAND, k2n512, level 1 (k N + 1 = 1025 words)", flush=True)
    out["cases"].append({"set": "k2n512", "level": 1, "op": "NAND", "key": key, "inputs": "level1", "operands": [0, 1],
                         "expected": gate2_level1("NAND", ins1[0], ins1[1], bk, ksk, K2N512)})
    bk, ksk, key = key_for(20261007, CGGI16)
    for op in ("NAND", "ORYN"):
        print(op, "cggi16 (n = 500, l = 2, Bg = 2^10)", flush=True)
        out["cases"].append({"set": "cggi16", "level": 0, "op": op, "key": key, "inputs": "level0_n500", "operands": [0, 1],
                             "expected": gate2(op, ins500[0], ins500[1], bk, ksk, CGGI16)})
    print("XOR, cggi16, level 1", flush=True)
    out["cases"].append({"set": "cggi16", "level": 1, "op": "XOR", "key": key, "inputs": "level1", "operands": [0, 1],
                         "expected": gate2_level1("XOR", ins1[0], ins1[1], bk, ksk, CGGI16)})
    # --- N = 2048 ring, 64-bit torus
    bk2, ksk2, key2 = key_for(20261005, LVL2)
    print("NAND, N = 2048", flush=True)
    out["cases"].append({"set": "lvl2", "level": 0, "op": "NAND", "key": key2, "inputs": "level0", "operands": [0, 1],
                         "expected": gate2("NAND", ins0[0], ins0[1], bk2, ksk2, LVL2)})
    out["seconds"] = round(time.time() - t0, 1)
    dst = os.path.join(HERE, "golden_independent_v5.json")
    json.dump(out, open(dst, "w"))
    print("wrote", dst, out["seconds"], "s")


def main_smallmod():
    """tests/golden/golden_independent_smallmod_v1.json: the small-modulus mode, a fixture of its own (the cases above keep their
    file and their words): NAND, XOR and MUX on level-0 ciphertexts, NAND on level-1 ciphertexts, NAND on the corner inputs `a`"""
    out = {"format": 5, "generator": "tests/golden/make_golden_independent.py smallmod (schoolbook mod P, no code shared with oracle/)", "cases": []}
    t0 = time.time()
    irng = np.random.default_rng(778)
    ins0 = [random_words(irng, 631) for _ in range(3)]
    ins1 = [random_words(irng, 1025) for _ in range(2)]
    edges = edge_inputs(irng)
    out["inputs"] = {"level0": [x.tolist() for x in ins0], "level1": [x.tolist() for x in ins1], "level0_edge_a": [x.tolist() for x in edges[0]]}
    # the switches at their corners (python integers against the array form)
    probe = np.array([0, 1, 3, 0x7FFFFFFF, 0x80000000, 0x80000001, 0xFFFFFFFE, 0xFFFFFFFF], np.uint32)
    assert [int(v) for v in torus32_to_ntt_mod(probe)] == [(int(a) * SMALL_P + 2**31) >> 32 for a in probe] and int(torus32_to_ntt_mod(probe)[-1]) == SMALL_P
    assert ntt_mod_to_torus32(0) == 0 and ntt_mod_to_torus32(SMALL_P - 1) == 0xFFFFFFF9 and abs(ntt_mod_to_torus32(SMALL_P // 2) - 2**31) < 8
    bk, ksk, key = key_for(20261009, SMALLMOD)
    R = SMALLMOD
    for op in ("NAND", "XOR"):
        print(op, "smallmod, level 0", flush=True)
        out["cases"].append({"set": "smallmod", "level": 0, "op": op, "key": key, "inputs": "level0", "operands": [0, 1],
                             "expected": gate2(op, ins0[0], ins0[1], bk, ksk, R)})
    print("MUX, smallmod, level 0", flush=True)
    out["cases"].append({"set": "smallmod", "level": 0, "op": "MUX", "key": key, "inputs": "level0", "operands": [0, 1, 2],
                         "expected": gate_mux(ins0[0], ins0[1], ins0[2], bk, ksk, R)})
    print("NAND, smallmod, level 1", flush=True)
    out["cases"].append({"set": "smallmod", "level": 1, "op": "NAND", "key": key, "inputs": "level1", "operands": [0, 1],
                         "expected": gate2_level1("NAND", ins1[0], ins1[1], bk, ksk, R)})
    print("NAND, smallmod, level 0, corner inputs a", flush=True)
    out["cases"].append({"set": "smallmod", "level": 0, "op": "NAND", "key": key, "inputs": "level0_edge_a", "operands": [0, 1],
                         "expected": gate2("NAND", edges[0][0], edges[0][1], bk, ksk, R)})
    out["seconds"] = round(time.time() - t0, 1)
    dst = os.path.join(HERE, "golden_independent_smallmod_v1.json")
    json.dump(out, open(dst, "w"))
    print("wrote", dst, out["seconds"], "s")


def cmux_trgsws(R, seed):
    """the two TRGSWs of the CMUXNTT cases, [(k+1) l][k+1][N] torus words flattened: uniform words, and extreme words with a first row
    of 0x80000000 (every word -2^31 in a signed reading)"""
    rng = np.random.default_rng(seed)
    K1, rows = R.k + 1, (R.k + 1) * R.l
    uniform = random_words(rng, rows * K1 * R.N)
    ext = np.array(EXTREME_WORDS, np.uint32)
    extreme = ext[rng.integers(0, ext.size, rows * K1 * R.N)]
    extreme[: K1 * R.N] = 0x80000000
    return uniform, extreme


def main_cmux_sets():
    """tests/golden/golden_independent_cmux_sets_v1.json (round 6): CMUXNTT on the OTHER compiled parameter sets -- in the reference the
    set TFHEpp selects at build time serves CMUXNTT / TRGSW2NTT too (src/bootstrap_gpu.cu:75-94,197-285 are templates over lvl1param) --
    as exact schoolbook sums: `k2n512` (three polynomials of 512, nine TRGSW rows) and `cggi16` (l = 2, Bg = 2^10: the set whose key the
    library takes in two 16-bit limbs; here there are no limbs, only the integer sum), each on a TRGSW of uniform words, on one of extreme
    words, and in place (res = c0 is the reference's usual call, test/test_cmux.cc).  A fixture of its own: the cases of v5 keep their
    file and their words."""
    out = {"format": 6, "generator": "tests/golden/make_golden_independent.py cmux_sets (schoolbook, no code shared with oracle/)", "cases": []}
    t0 = time.time()
    irng = np.random.default_rng(779)
    out["inputs"] = {}
    for name, R in (("k2n512", K2N512), ("cggi16", CGGI16)):
        K1, rows = R.k + 1, (R.k + 1) * R.l
        trl = [random_words(irng, K1 * R.N) for _ in range(2)]
        trgsw, trgsw_x = cmux_trgsws(R, 20261010 + R.N + R.l)
        trl[0][:8] = trl[1][:8]                               # difference 0 -> the digits of the bare offset
        out["inputs"]["trlwe_" + name] = [x.tolist() for x in trl]
        # like the keys: regenerated from the seed by the test (cmux_trgsws below), pinned by a hash
        out["inputs"]["trgsw_" + name] = {"seed": 20261010 + R.N + R.l, "sha256": [hashlib.sha256(t.tobytes()).hexdigest() for t in (trgsw, trgsw_x)]}
        as_trlwe = lambda x, R=R, K1=K1: [[int(v) for v in x[j * R.N:(j + 1) * R.N]] for j in range(K1)]
        flat = lambda t: [int(v) for comp in t for v in comp]
        for which, tg in (("uniform", trgsw), ("extreme", trgsw_x)):
            print("CMUXNTT", name, which, flush=True)
            out["cases"].append({"set": name, "level": 2, "op": "CMUXNTT", "trgsw": ["uniform", "extreme"].index(which), "inputs": "trlwe_" + name,
                                 "operands": [0, 1], "expected": flat(cmux(tg.reshape(rows, K1, R.N), as_trlwe(trl[0]), as_trlwe(trl[1]), R))})
        # chained: the second CMUX takes the first one's result as c0 and the operands swapped
        first = cmux(trgsw.reshape(rows, K1, R.N), as_trlwe(trl[0]), as_trlwe(trl[1]), R)
        out["cases"].append({"set": name, "level": 2, "op": "CMUXNTT_CHAINED", "trgsw": 1, "inputs": "trlwe_" + name, "operands": [1, 0],
                             "expected": flat(cmux(trgsw_x.reshape(rows, K1, R.N), as_trlwe(trl[1]), first, R))})
    out["seconds"] = round(time.time() - t0, 1)
    dst = os.path.join(HERE, "golden_independent_cmux_sets_v1.json")
    json.dump(out, open(dst, "w"))
    print("wrote", dst, out["seconds"], "s")


if __name__ == "__main__":
    if sys.argv[1:] == ["cmux_sets"]:
        main_cmux_sets()
    elif sys.argv[1:] == ["smallmod"]:
        main_smallmod()
    else:
        main()
