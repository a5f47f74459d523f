"""ctypes wrapper of oracle/liboracle.so (CPU restatement, test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libplain_ref.so")

n, N = 630, 1024
LVL_WORDS = (n + 1, N + 1)
BK_WORDS = n * 6 * 2 * N
KSK_WORDS = N * 8 * 2 * (n + 1)
MU = 1 << 29
OPS = ["NAND", "NOR", "XNOR", "AND", "OR", "XOR", "ANDNY", "ANDYN", "ORNY", "ORYN", "MUX", "NMUX", "NOT", "COPY"]

_u32 = np.ctypeslib.ndpointer(np.uint32, flags="C")
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C")
_u8 = np.ctypeslib.ndpointer(np.uint8, flags="C")
_u64 = np.ctypeslib.ndpointer(np.uint64, flags="C")


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("tfhe_oracle.c", "tfhe_oracle.h", "tfhe_oracle_lvl2.c", "tfhe_oracle_lvl2.h", "cpu_fast.c")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    if not os.path.exists(REF_LIB) and os.path.exists("/root/reference/test/plain.h"):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)


SETS = ("default", "k2n512", "cggi16", "smallmod")       # parameter sets the oracle is compiled for (oracle/tfhe_oracle.h)


def load_set(name):
    """The oracle compiled for a named parameter set (liboracle_<name>.so; "default" = liboracle.so)."""
    if name == "default":
        return load()
    path = os.path.join(ORACLE_DIR, f"liboracle_{name}.so")
    src = os.path.join(ORACLE_DIR, "tfhe_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, os.path.basename(path)], stdout=subprocess.DEVNULL)
    return _bind_gate_path(ctypes.CDLL(path))


def _bind_gate_path(L):
    L.orc_get_params.restype = ctypes.c_char_p
    L.orc_get_params.argtypes = [ctypes.POINTER(ctypes.c_int)]
    L.orc_keygen.argtypes = [ctypes.c_uint64, _u32, _u32]
    L.orc_bkgen.argtypes = [ctypes.c_uint64, _u32, _u32, _u32]
    L.orc_kskgen.argtypes = [ctypes.c_uint64, _u32, _u32, _u32]
    L.orc_tlwe_encrypt_batch.argtypes = [ctypes.c_uint64, ctypes.c_int, _u32, _u8, ctypes.c_size_t, _u32]
    L.orc_tlwe_decrypt_batch.argtypes = [ctypes.c_int, _u32, _u32, ctypes.c_size_t, _u8]
    L.orc_evalkey_create.restype = ctypes.c_void_p
    L.orc_evalkey_create.argtypes = [_u32, _u32]
    L.orc_evalkey_destroy.argtypes = [ctypes.c_void_p]
    L.orc_blind_rotate.argtypes = [ctypes.c_void_p, _u32, _u32, ctypes.c_int]
    L.orc_sample_extract0.argtypes = [_u32, _u32]
    L.orc_keyswitch.argtypes = [ctypes.c_void_p, _u32, _u32]
    L.orc_gate_batch.argtypes = [ctypes.c_void_p, _i32, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                                 _u32, _u32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.orc_cmux.argtypes = [_u32, _u32, _u32, _u32]
    L.orc_truth.argtypes = [ctypes.c_int] * 4
    return L


def set_params(L):
    """(name, dict) of the parameter set a loaded oracle library was compiled for."""
    v = (ctypes.c_int * 7)()
    L.orc_get_params.restype = ctypes.c_char_p
    name = L.orc_get_params(v).decode()
    d = dict(zip(("n", "Nbit", "k", "l", "Bgbit", "t", "basebit"), list(v)))
    d["N"] = 1 << d["Nbit"]
    return name, d


def load():
    build()
    L = ctypes.CDLL(LIB)
    L.orc_keygen.argtypes = [ctypes.c_uint64, _u32, _u32]
    L.orc_bkgen.argtypes = [ctypes.c_uint64, _u32, _u32, _u32]
    L.orc_kskgen.argtypes = [ctypes.c_uint64, _u32, _u32, _u32]
    L.orc_tlwe_encrypt_batch.argtypes = [ctypes.c_uint64, ctypes.c_int, _u32, _u8, ctypes.c_size_t, _u32]
    L.orc_tlwe_decrypt_batch.argtypes = [ctypes.c_int, _u32, _u32, ctypes.c_size_t, _u8]
    L.orc_polymul_schoolbook.argtypes = [_u32, _i32, _u32]
    L.orc_polymul_ntt.argtypes = [_u32, _i32, _u32]
    L.orc_ntt_forward.argtypes = [_u64]
    L.orc_ntt_inverse.argtypes = [_u64]
    for f in ("orc_ntt_modulus", "orc_ntt_psi", "orc_ntt_barrett_mu", "orc_ntt_n_inverse"):
        getattr(L, f).restype = ctypes.c_uint64
    L.orc_ntt_mulmod.restype = ctypes.c_uint64
    L.orc_ntt_mulmod.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    L.orc_evalkey_create.restype = ctypes.c_void_p
    L.orc_evalkey_create.argtypes = [_u32, _u32]
    L.orc_evalkey_destroy.argtypes = [ctypes.c_void_p]
    L.orc_blind_rotate.argtypes = [ctypes.c_void_p, _u32, _u32, ctypes.c_int]
    L.orc_sample_extract0.argtypes = [_u32, _u32]
    L.orc_keyswitch.argtypes = [ctypes.c_void_p, _u32, _u32]
    L.orc_gate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _u32, _u32, _u32, _u32]
    L.orc_gate_batch.argtypes = [ctypes.c_void_p, _i32, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                                 _u32, _u32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.orc_cmux.argtypes = [_u32, _u32, _u32, _u32]
    L.orc_sample_extract_keyswitch.argtypes = [ctypes.c_void_p, _u32, _u32]
    L.orc_refresh.argtypes = [ctypes.c_void_p, _u32, _u32]
    L.orc_truth.argtypes = [ctypes.c_int] * 4
    L.orc_gate_coeffs.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 3
    # optimised CPU baseline (oracle/cpu_fast.c): same words as orc_gate_batch at level 0
    L.fast_evalkey_create.restype = ctypes.c_void_p
    L.fast_evalkey_create.argtypes = [_u32, _u32]
    L.fast_evalkey_destroy.argtypes = [ctypes.c_void_p]
    L.fast_gate_batch.argtypes = [ctypes.c_void_p, _i32, ctypes.c_int, ctypes.c_size_t, _u32, _u32, _u32, ctypes.c_int]
    # N = 2048 / 64-bit torus (oracle/tfhe_oracle_lvl2.h)
    L.orc2_keygen.argtypes = [ctypes.c_uint64, _u32]
    L.orc2_bkgen.argtypes = [ctypes.c_uint64, _u32, _u32, _u64]
    L.orc2_kskgen.argtypes = [ctypes.c_uint64, _u32, _u32, _u32]
    L.orc2_polymul_schoolbook.argtypes = [_u64, _i32, _u64]
    L.orc2_polymul_ntt.argtypes = [_u64, _i32, _u64]
    L.orc2_tlwe_phase.restype = ctypes.c_uint64
    L.orc2_tlwe_phase.argtypes = [_u32, _u64]
    L.orc2_evalkey_create.restype = ctypes.c_void_p
    L.orc2_evalkey_create.argtypes = [_u64, _u32]
    L.orc2_evalkey_destroy.argtypes = [ctypes.c_void_p]
    L.orc2_blind_rotate.argtypes = [ctypes.c_void_p, _u64, _u32, ctypes.c_int]
    L.orc2_sample_extract0.argtypes = [_u64, _u64]
    L.orc2_keyswitch.argtypes = [ctypes.c_void_p, _u32, _u64]
    L.orc2_gate_batch.argtypes = [ctypes.c_void_p, _i32, ctypes.c_int, ctypes.c_size_t,
                                  _u32, _u32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return L


class Keys:
    """Seeded secret keys + evaluation key (deterministic: same seed, same words), for the parameter
    set the library `L` was compiled for."""

    def __init__(self, L, seed=1):
        self.L = L
        self.seed = seed
        self.set_name, p = set_params(L)
        self.n, self.N, self.k = p["n"], p["N"], p["k"]
        self.words = (self.n + 1, self.k * self.N + 1)
        self.bk_words = self.n * (self.k + 1) * p["l"] * (self.k + 1) * self.N
        self.ksk_words = self.k * self.N * p["t"] * (1 << (p["basebit"] - 1)) * (self.n + 1)
        self.s0 = np.zeros(self.n, np.uint32)
        self.s1 = np.zeros(self.k * self.N, np.uint32)
        L.orc_keygen(seed, self.s0, self.s1)
        self.bk = np.zeros(self.bk_words, np.uint32)
        self.ksk = np.zeros(self.ksk_words, np.uint32)
        L.orc_bkgen(seed + 1000, self.s0, self.s1, self.bk)
        L.orc_kskgen(seed + 2000, self.s0, self.s1, self.ksk)
        self.ek = L.orc_evalkey_create(self.bk, self.ksk)

    def key(self, level):
        return self.s1 if level else self.s0

    def encrypt(self, bits, level, seed):
        bits = np.ascontiguousarray(bits, dtype=np.uint8).ravel()
        cts = np.zeros(bits.size * self.words[level], np.uint32)
        self.L.orc_tlwe_encrypt_batch(seed, level, self.key(level), bits, bits.size, cts)
        return cts.reshape(bits.size, self.words[level])

    def decrypt(self, cts, level):
        cts = np.ascontiguousarray(cts, dtype=np.uint32).reshape(-1, self.words[level])
        bits = np.zeros(cts.shape[0], np.uint8)
        self.L.orc_tlwe_decrypt_batch(level, self.key(level), cts.ravel(), cts.shape[0], bits)
        return bits

    def gate_batch(self, ops, level, in0, in1=None, in2=None, threads=None):
        in0 = np.ascontiguousarray(in0, dtype=np.uint32)
        count = in0.reshape(-1, self.words[level]).shape[0]
        if np.isscalar(ops):
            ops_arr, stride = np.array([ops], np.int32), 0
        else:
            ops_arr, stride = np.ascontiguousarray(ops, np.int32), 1
        out = np.zeros(count * self.words[level], np.uint32)
        p1 = p2 = None
        if in1 is not None:
            in1 = np.ascontiguousarray(in1, np.uint32); p1 = in1.ctypes.data
        if in2 is not None:
            in2 = np.ascontiguousarray(in2, np.uint32); p2 = in2.ctypes.data
        if threads is None:
            threads = self.L.orc_max_threads()
        self.L.orc_gate_batch(self.ek, ops_arr, stride, level, count, out, in0.ravel(), p1, p2, threads)
        return out.reshape(count, self.words[level])

    def blind_rotate(self, tlwe0, steps=-1):
        acc = np.zeros((self.k + 1) * self.N, np.uint32)
        self.L.orc_blind_rotate(self.ek, acc, np.ascontiguousarray(tlwe0, np.uint32), steps)
        return acc

    def keyswitch(self, tlwe1):
        out = np.zeros(self.n + 1, np.uint32)
        self.L.orc_keyswitch(self.ek, out, np.ascontiguousarray(tlwe1, np.uint32))
        return out


N2 = 2048
LVL2_WORDS = N2 + 1
BK2_WORDS = n * 8 * 2 * N2          # uint64
KSK2_WORDS = N2 * 7 * 2 * (n + 1)   # uint32
MU2 = 1 << 61


class KeysLvl2:
    """lvl02 bootstrapping key and lvl20 key-switching key over the SAME lvl0 key as `base`."""

    def __init__(self, L, base, seed=1):
        self.L, self.base = L, base
        self.s0 = base.s0
        self.s2 = np.zeros(N2, np.uint32)
        L.orc2_keygen(seed, self.s2)
        self.bk = np.zeros(BK2_WORDS, np.uint64)
        self.ksk = np.zeros(KSK2_WORDS, np.uint32)
        L.orc2_bkgen(seed + 3000, self.s0, self.s2, self.bk)
        L.orc2_kskgen(seed + 4000, self.s0, self.s2, self.ksk)
        self.ek = L.orc2_evalkey_create(self.bk, self.ksk)

    def blind_rotate(self, tlwe0, steps=-1):
        acc = np.zeros(2 * N2, np.uint64)
        self.L.orc2_blind_rotate(self.ek, acc, np.ascontiguousarray(tlwe0, np.uint32), steps)
        return acc

    def sample_extract(self, acc):
        out = np.zeros(LVL2_WORDS, np.uint64)
        self.L.orc2_sample_extract0(out, np.ascontiguousarray(acc, np.uint64))
        return out

    def keyswitch(self, tlwe2):
        out = np.zeros(n + 1, np.uint32)
        self.L.orc2_keyswitch(self.ek, out, np.ascontiguousarray(tlwe2, np.uint64))
        return out

    def phase2(self, tlwe2):
        return self.L.orc2_tlwe_phase(self.s2, np.ascontiguousarray(tlwe2, np.uint64))

    def gate_batch(self, ops, in0, in1=None, in2=None, threads=None):
        in0 = np.ascontiguousarray(in0, dtype=np.uint32)
        count = in0.reshape(-1, n + 1).shape[0]
        if np.isscalar(ops):
            ops_arr, stride = np.array([ops], np.int32), 0
        else:
            ops_arr, stride = np.ascontiguousarray(ops, np.int32), 1
        out = np.zeros(count * (n + 1), np.uint32)
        p1 = p2 = None
        if in1 is not None:
            in1 = np.ascontiguousarray(in1, np.uint32); p1 = in1.ctypes.data
        if in2 is not None:
            in2 = np.ascontiguousarray(in2, np.uint32); p2 = in2.ctypes.data
        if threads is None:
            threads = self.L.orc_max_threads()
        self.L.orc2_gate_batch(self.ek, ops_arr, stride, count, out, in0.ravel(), p1, p2, threads)
        return out.reshape(count, n + 1)


REF_NAMES = {   # mangled names of namespace cufhe's truth functions in /root/reference/test/plain.h:10-69
    "NAND": "_ZN5cufhe9NandCheckERhRKhS2_", "OR": "_ZN5cufhe7OrCheckERhRKhS2_",
    "ORYN": "_ZN5cufhe9OrYNCheckERhRKhS2_", "ORNY": "_ZN5cufhe9OrNYCheckERhRKhS2_",
    "AND": "_ZN5cufhe8AndCheckERhRKhS2_", "ANDYN": "_ZN5cufhe10AndYNCheckERhRKhS2_",
    "ANDNY": "_ZN5cufhe10AndNYCheckERhRKhS2_", "XOR": "_ZN5cufhe8XorCheckERhRKhS2_",
    "XNOR": "_ZN5cufhe9XnorCheckERhRKhS2_", "MUX": "_ZN5cufhe8MuxCheckERhRKhS2_S2_",
    "NMUX": "_ZN5cufhe9NMuxCheckERhRKhS2_S2_", "NOT": "_ZN5cufhe8NotCheckERhRKh",
    "COPY": "_ZN5cufhe9CopyCheckERhRKh",
}
_REF = None


def ref_plain():
    """oracle/_ref/libplain_ref.so -- the reference's own truth functions (test/plain.h compiled where it lies by
    oracle/Makefile; the built file travels to the GPU box) -- or None when it was never built."""
    global _REF
    if _REF is None:
        _REF = ctypes.CDLL(REF_LIB) if os.path.exists(REF_LIB) else False
    return _REF or None


def ref_truth(name, a, b=0, c=0):
    """truth value of gate `name` from the reference's plain.h; None if the gate is not in plain.h (NOR) or _ref is absent"""
    ref = ref_plain()
    if ref is None or name not in REF_NAMES:
        return None
    fn = getattr(ref, REF_NAMES[name])
    u8 = ctypes.c_uint8
    out, x, y, z = u8(7), u8(int(a)), u8(int(b)), u8(int(c))
    if name in ("NOT", "COPY"):
        fn(ctypes.byref(out), ctypes.byref(x))
    elif name in ("MUX", "NMUX"):
        fn(ctypes.byref(out), ctypes.byref(x), ctypes.byref(y), ctypes.byref(z))
    else:
        fn(ctypes.byref(out), ctypes.byref(x), ctypes.byref(y))
    return int(out.value)


def truth(L, op, a, b=0, c=0):
    """The expected plaintext of a gate: from the REFERENCE's truth functions (test/plain.h through oracle/_ref) whenever that
    library exists -- on the GPU box too -- and from the oracle's own table only for what plain.h lacks (NOR) or when _ref
    was never built.  tests/test_oracle.py pins the two against each other."""
    r = ref_truth(OPS[op], a, b, c)
    return r if r is not None else L.orc_truth(op, int(a), int(b), int(c))
