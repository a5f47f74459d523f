"""CPU tests of the FP64 prime-field arithmetic the HIP kernels use (cufhe_amd/csrc/fpfield.h)
and of the lazy-reduction schedule of cufhe_amd/csrc/ntt_wave.h, replayed on the host by
tests/host/host_model.cpp.  The product path is NOT exercised here (no GPU): this checks the
exactness argument -- every intermediate an exact integer < 2^53, every multiplication
input inside its documented range -- on random and worst-case inputs, and the final words
against the oracle's schoolbook product."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host", "host_model.cpp")
LIB = os.path.join(HERE, "host", "libhost_model.so")
P = 875781160960001


@pytest.fixture(scope="module")
def hm():
    deps = [SRC, os.path.join(ol.ROOT, "cufhe_amd", "csrc", "fpfield.h"), os.path.join(ol.ROOT, "cufhe_amd", "csrc", "ntt_r4.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", LIB, SRC])
    L = ctypes.CDLL(LIB)
    f64 = np.ctypeslib.ndpointer(np.float64, flags="C")
    L.hm_check_mulmod.argtypes = [f64, f64, ctypes.c_int, ctypes.c_int]
    L.hm_check_reduce_lift.argtypes = [f64, ctypes.c_int]
    u32 = np.ctypeslib.ndpointer(np.uint32, flags="C")
    i32 = np.ctypeslib.ndpointer(np.int32, flags="C")
    L.hm_polymul.argtypes = [u32, i32, u32, f64]
    L.hm_external_product.argtypes = [u32, i32, u32, f64]
    L.hm_external_product_split.argtypes = [u32, i32, u32, f64]
    L.hm_external_product_r4.argtypes = [u32, i32, u32, f64]
    L.hm_check_mul_root4.argtypes = [f64, ctypes.c_int]
    L.hm_check_mulmod_add.argtypes = [f64, f64, f64, ctypes.c_int, ctypes.c_int]
    L.hm_p.restype = ctypes.c_double
    return L


def test_prime_and_root():
    import sympy
    assert sympy.isprime(P) and (P - 1) % (1 << 12) == 0 and P == 5440**4 + 1
    psi = 423584205157050
    assert pow(psi, 1024, P) == P - 1 and pow(psi, 2048, P) == 1
    assert pow(psi, 256, P) == 5440 and pow(psi, 512, P) == 5440**2     # small 8th and 4th roots
    # exactness bound: (k+1) l N (Bg/2) 2^31 < p/2   (signed BK words)
    assert 6144 * 32 * 2**31 < P // 2
    assert float.fromhex("0x1.491cc17c934a8p-50") == 1.0 / P


def test_header_constants_match(hm):
    assert hm.hm_p() == float(P)


@pytest.mark.parametrize("wide", [0, 1])
def test_mulmod_exact_and_bounded(hm, wide):
    rng = np.random.default_rng(1 + wide)
    count = 200000
    lim = (2**53 if wide else 2**52) - 1
    a = rng.integers(-lim, lim + 1, size=count).astype(np.float64)
    w = rng.integers(-(P // 2), P // 2 + 1, size=count).astype(np.float64)
    # edges: largest legal inputs, largest twiddle, zeros, ones
    a[:8] = [lim, -lim, lim, -lim, 0, 1, -1, lim]
    w[:8] = [P // 2, P // 2, -(P // 2), -(P // 2), P // 2, P // 2, -(P // 2), 1]
    assert hm.hm_check_mulmod(a, w, count, wide) == 0


def test_reduce_and_lift(hm):
    rng = np.random.default_rng(3)
    a = rng.integers(-(2**53 - 1), 2**53, size=200000).astype(np.float64)
    a[:6] = [0, 1, -1, P // 2, -(P // 2), 2**53 - 1]
    assert hm.hm_check_reduce_lift(a, a.size) == 0


def _school(oracle, a, b):
    want = np.zeros(ol.N, np.uint32)
    oracle.orc_polymul_schoolbook(want, np.ascontiguousarray(a, np.int32), np.ascontiguousarray(b, np.uint32))
    return want


def test_polymul_schedule_random_and_bounds(hm, oracle):
    rng = np.random.default_rng(5)
    fwd_doc = [.0061, .0061, .51, 1.06, 1.66, 2.32, 3.05, 3.85, 4.72, 6.18]       # ntt_wave.h header (digits in): stages 0-1 are one exact radix-4 butterfly
    inv_doc = [1.0, 2.0, 4.0, 8.0, 1.0, 2.0, 4.0, 8.0, 1.0, 2.0]
    for trial in range(20):
        a = rng.integers(-32, 32, size=ol.N, dtype=np.int32)
        b = rng.integers(0, 2**32, size=ol.N, dtype=np.uint64).astype(np.uint32)
        res = np.zeros(ol.N, np.uint32)
        st = np.zeros(24)
        hm.hm_polymul(res, a, b, st)
        assert st[0] == 0, "non-integer or out-of-range intermediate"
        assert st[1] < 10.285 and st[2] < 5.142 and st[3] < 10.285
        assert all(st[4 + s] <= fwd_doc[s] + 1e-9 for s in range(10)), st[4:14]
        assert all(st[14 + s] <= inv_doc[s] + 1e-9 for s in range(10)), st[14:24]
        assert np.array_equal(res, _school(oracle, a, b))


def test_exact_radix4_first_stages_up_to_bg_1024(hm, oracle):
    """The first two forward stages are one exact radix-4 butterfly on the inputs (ntt_wave.h: ct_four_stages<SMALL_IN>, zeta^3 I = -zeta).
    It stays exact up to digits of Bg = 2^10 (the cggi16 set, kernels_ps.hip.h: |d| (1 + I + zeta + zeta^3) = 2^46.2): every intermediate an
    integer below 2^53, the identity checked in integer arithmetic by the host model, products == schoolbook -- including the extreme digits."""
    rng = np.random.default_rng(11)
    for trial in range(6):
        a = rng.integers(-512, 512, size=ol.N, dtype=np.int32)
        if trial == 0:
            a[:] = -512
        if trial == 1:
            a[:] = 511
            a[::3] = -512
        # the key operand as that set takes it: balanced 16-bit limbs (a whole 32-bit word against Bg = 2^10 digits leaves the exact range)
        b = rng.integers(-2**15, 2**15, size=ol.N, dtype=np.int64).astype(np.uint32)
        if trial < 2:
            b[:] = np.uint32(2**32 - 2**15)
        res = np.zeros(ol.N, np.uint32)
        st = np.zeros(24)
        hm.hm_polymul(res, a, b, st)
        assert st[0] == 0, "non-integer or out-of-range intermediate"
        assert st[1] < 10.285 and st[2] < 5.142 and st[3] < 10.285
        assert st[4] <= 512 * (1 + 29593600 + 5440 + 5440**3) / P + 1e-12 and st[13] < 6.4      # 0.094 p after the butterfly, spectrum below 6.4 p
        assert np.array_equal(res, _school(oracle, a, b))


def test_external_product_worst_case(hm, oracle):
    """All digits -32 against all BK words 0x80000000: coefficient N-1 of every row reaches
    1024*32*2^31, and the six rows add up to the bound 2^48.585 -- still exact."""
    dig = np.full((6, ol.N), -32, np.int32)
    bk = np.full((6, 2, ol.N), 0x80000000, np.uint32)
    out = np.zeros(2 * ol.N, np.uint32)
    st = np.zeros(4)
    hm.hm_external_product(out, dig.ravel(), bk.ravel(), st)
    assert st[0] == 0 and st[1] < 10.285 and st[2] < 5.142 and st[3] < 10.285
    want = np.zeros(ol.N, np.uint32)
    for row in range(6):
        want += _school(oracle, dig[row], bk[row, 0])
    assert np.array_equal(out[:ol.N], want) and np.array_equal(out[ol.N:], want)
    # the true integer at coefficient N-1 is +6*2^46 (not representable mod 2^32 without care)
    assert int(want[ol.N - 1]) == (6 * 2**46) % 2**32


def test_external_product_random(hm, oracle):
    rng = np.random.default_rng(9)
    for trial in range(5):
        dig = rng.integers(-32, 32, size=(6, ol.N), dtype=np.int32)
        bk = rng.integers(0, 2**32, size=(6, 2, ol.N), dtype=np.uint64).astype(np.uint32)
        out = np.zeros(2 * ol.N, np.uint32)
        st = np.zeros(4)
        hm.hm_external_product(out, dig.ravel(), bk.ravel(), st)
        assert st[0] == 0 and st[1] < 10.285
        for c in range(2):
            want = np.zeros(ol.N, np.uint32)
            for row in range(6):
                want += _school(oracle, dig[row], bk[row, c])
            assert np.array_equal(out[c * ol.N:(c + 1) * ol.N], want)


def test_split_transform_schedule(hm, oracle):
    """The schedule of the low-latency kernel (two 512-point halves per transform,
    ntt_wave512.h): same words as the plain schedule, every intermediate an exact integer inside
    its documented range -- for the worst case and for random inputs."""
    rng = np.random.default_rng(10)
    cases = [(np.full((6, ol.N), -32, np.int32), np.full((6, 2, ol.N), 0x80000000, np.uint32))]
    for _ in range(3):
        cases.append((rng.integers(-32, 32, size=(6, ol.N), dtype=np.int32),
                      rng.integers(0, 2**32, size=(6, 2, ol.N), dtype=np.uint64).astype(np.uint32)))
    for dig, bk in cases:
        out = np.zeros(2 * ol.N, np.uint32)
        ref = np.zeros(2 * ol.N, np.uint32)
        st = np.zeros(4)
        hm.hm_external_product_split(out, dig.ravel(), bk.ravel(), st)
        hm.hm_external_product(ref, dig.ravel(), bk.ravel(), np.zeros(4))
        assert st[0] == 0 and st[1] < 10.285 and st[2] < 5.142 and st[3] < 10.285, st
        assert np.array_equal(out, ref)
        want = np.zeros(ol.N, np.uint32)
        for row in range(6):
            want += _school(oracle, dig[row], bk[row, 0])
        assert np.array_equal(out[:ol.N], want)


def test_mul_root4_exact_and_bounded(hm):
    """fpf::mul_root4: I x mod p in four operations (p = I^2 + 1), any |x| < 2^53 in, |result| <= (0.5 + 5e-7) p."""
    rng = np.random.default_rng(21)
    a = rng.integers(-(2**53 - 1), 2**53, size=200000).astype(np.float64)
    I = 29593600
    a[:12] = [0, 1, -1, I, -I, I // 2, I // 2 + 1, -(I // 2) - 1, 2**53 - 1, -(2**53 - 1), P // 2, -(P // 2)]
    # ties of the quotient rounding: x = (2k + 1) I / 2
    a[12:20] = [(2 * k + 1) * I // 2 for k in (0, 1, 5, 1000, 2**27, -1, -6, -2**27)]
    assert hm.hm_check_mul_root4(a, a.size) == 0


@pytest.mark.parametrize("wide", [0, 1])
def test_mulmod_add_exact_and_bounded(hm, wide):
    """fpf::mulmod_add(_wide): a w + c reduced in one step; c up to what five earlier products can have left."""
    rng = np.random.default_rng(23 + wide)
    count = 200000
    lim = (2**53 if wide else 2**52) - 1
    a = rng.integers(-lim, lim + 1, size=count).astype(np.float64)
    w = rng.integers(-(P // 2), P // 2 + 1, size=count).astype(np.float64)
    cmax = int((10.285 - (1.0 if wide else 0.5) - 0.14585 * (lim / P) - 0.01) * P)
    c = rng.integers(-cmax, cmax + 1, size=count).astype(np.float64)
    a[:8] = [lim, -lim, lim, -lim, 0, 1, -1, lim]
    w[:8] = [P // 2, P // 2, -(P // 2), -(P // 2), P // 2, P // 2, -(P // 2), 1]
    c[:8] = [cmax, cmax, -cmax, cmax, cmax, -cmax, 0, cmax]
    assert hm.hm_check_mulmod_add(a, w, c, count, wide) == 0


def test_radix4_schedule_worst_case_and_random(hm, oracle):
    """The schedule of blind_rotate_kernel (ntt_r4.h: radix-4 passes, per-register compile-time bounds, reductions only on the
    registers that need one, the last product of a sum reducing it), run by the device's own pass functions on 64 emulated lanes:
    every value an integer below 2^53 and below the bound of its register, words == the radix-2 schedule == schoolbook --
    for the worst case (all digits -32 against all key words 0x80000000: the sum reaches the exactness bound) and random inputs."""
    rng = np.random.default_rng(12)
    cases = [(np.full((6, ol.N), -32, np.int32), np.full((6, 2, ol.N), 0x80000000, np.uint32)),
             (np.full((6, ol.N), 31, np.int32), np.full((6, 2, ol.N), 0x7fffffff, np.uint32))]
    alt = np.full((6, ol.N), -32, np.int32)
    alt[:, ::2] = 31
    cases.append((alt, np.full((6, 2, ol.N), 0x80000000, np.uint32)))
    for _ in range(4):
        cases.append((rng.integers(-32, 32, size=(6, ol.N), dtype=np.int32),
                      rng.integers(0, 2**32, size=(6, 2, ol.N), dtype=np.uint64).astype(np.uint32)))
    for dig, bk in cases:
        out = np.zeros(2 * ol.N, np.uint32)
        ref = np.zeros(2 * ol.N, np.uint32)
        st = np.zeros(40)
        hm.hm_external_product_r4(out, dig.ravel(), bk.ravel(), st)
        hm.hm_external_product(ref, dig.ravel(), bk.ravel(), np.zeros(4))
        assert st[0] == 0 and st[1] < 10.285 and st[4] <= 1.0, st[:5]
        assert st[5:21].max() < 5.3 and st[21:37].max() < 10.285       # spectrum bound per register; inverse output bound
        assert np.array_equal(out, ref)
        want = np.zeros(ol.N, np.uint32)
        for row in range(6):
            want += _school(oracle, dig[row], bk[row, 0])
        assert np.array_equal(out[:ol.N], want)
