"""Python mirror of the cuFHE host API for the gate path, over the C ABI.

Names, argument order and completion semantics follow include/cufhe_gpu.cuh of the
reference (SetGPUNum / Initialize / CleanUp / Synchronize / Stream / StreamQuery / Ctxt /
And ... NMux, Not, Copy and the g-prefixed device-resident variants,
/root/reference/include/cufhe_gpu.cuh:54-313, src/cufhe_gates_gpu.cu:148-665).
Everything here is plumbing: the arithmetic runs in libcufhe_amd.so.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import lib, check, Params, Profile, Lvl2Params, SchedStats, PsParams

# op codes (include/cufhe_amd.h)
NAND, NOR, XNOR, AND, OR, XOR, ANDNY, ANDYN, ORNY, ORYN, MUX, NMUX, NOT, COPY = range(14)
OP_NAMES = ["NAND", "NOR", "XNOR", "AND", "OR", "XOR", "ANDNY", "ANDYN", "ORNY", "ORYN",
            "MUX", "NMUX", "NOT", "COPY"]


def params():
    p = Params()
    check(lib.cufhe_amd_get_params(ctypes.byref(p)))
    return p


PARAMS = params()
LVL_WORDS = (PARAMS.lvl0_words, PARAMS.lvl1_words)

_stream_count = 0          # `streamCount`, src/cufhe_gates_gpu.cu:36


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if isinstance(a, np.ndarray) else a


def SetGPUNum(gpu_num):
    check(lib.cufhe_amd_set_gpu_num(int(gpu_num)))


def GetGPUNum():
    return lib.cufhe_amd_get_gpu_num()


def DeviceCount():
    return lib.cufhe_amd_device_count()


def device_identity(device=0):
    """Which physical GPU a logical device is: {'pci': ..., 'uuid': ..., 'hip_device': ..., 'local_cpus': ...}."""
    buf = ctypes.create_string_buffer(512)
    check(lib.cufhe_amd_device_identity(int(device), buf, len(buf)))
    return dict(kv.split("=", 1) for kv in buf.value.decode().split())


def device_cus(device=0):
    """compute units of a logical device ("cus_override" included): the unit of the launch-shape and flush rules"""
    return check(lib.cufhe_amd_device_cus(int(device)))


def device_mem_info(device=0):
    """(free, total) bytes of the device (hipMemGetInfo)"""
    f, t = ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(lib.cufhe_amd_device_mem_info(int(device), ctypes.byref(f), ctypes.byref(t)))
    return f.value, t.value


def Initialize(bk=None, ksk=None):
    """Initialize() / Initialize(ek): bk, ksk are the torus-domain keys as uint32 arrays."""
    if bk is None:
        check(lib.cufhe_amd_initialize_ntt())
        return
    bk = np.ascontiguousarray(bk, dtype=np.uint32).ravel()
    ksk = np.ascontiguousarray(ksk, dtype=np.uint32).ravel()
    check(lib.cufhe_amd_initialize(_ptr(bk), bk.size, _ptr(ksk), ksk.size))


def CleanUp():
    check(lib.cufhe_amd_cleanup())


def Synchronize():
    check(lib.cufhe_amd_synchronize())


class Stream:
    """class Stream, include/cufhe_gpu.cuh:152-189: default ctor round-robins devices,
    Create() makes a non-blocking stream, the destructor does not destroy it."""

    def __init__(self, device_id=None):
        global _stream_count
        self._device_id = (_stream_count % GetGPUNum()) if device_id is None else int(device_id)
        _stream_count += 1
        self._st = ctypes.c_void_p(None)

    def Create(self):
        check(lib.cufhe_amd_stream_create(self._device_id, ctypes.byref(self._st)))

    def Destroy(self):
        check(lib.cufhe_amd_stream_destroy(self._device_id, self._st))
        self._st = ctypes.c_void_p(None)

    def st(self):
        return self._st

    def device_id(self):
        return self._device_id


def StreamQuery(st):
    return check(lib.cufhe_amd_stream_query(st.device_id(), st.st())) == 1


class DeviceBuffer:
    """`words` uint32 words of device memory on one GPU."""

    def __init__(self, words, device=0):
        self.words, self.device = int(words), int(device)
        p = ctypes.c_void_p()
        check(lib.cufhe_amd_malloc(self.device, self.words * 4, ctypes.byref(p)))
        self.ptr = p.value
        _lib.live.add(self)

    def upload(self, host, stream=None):
        host = np.ascontiguousarray(host, dtype=np.uint32).ravel()
        assert host.size <= self.words
        check(lib.cufhe_amd_memcpy_h2d(self.device, stream, self.ptr, _ptr(host), host.size * 4))
        check(lib.cufhe_amd_stream_synchronize(self.device, stream))
        return self

    def download(self, words=None, stream=None):
        out = np.empty(self.words if words is None else words, dtype=np.uint32)
        check(lib.cufhe_amd_memcpy_d2h(self.device, stream, _ptr(out), self.ptr, out.size * 4))
        check(lib.cufhe_amd_stream_synchronize(self.device, stream))
        return out

    def free(self):
        if self.ptr and not _lib.closed:
            check(lib.cufhe_amd_free(self.device, self.ptr))
        self.ptr = None

    release = free

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Ctxt:
    """template<class P> struct Ctxt, include/cufhe_gpu.cuh:102-121: a pinned host TLWE
    (`tlwehost`) plus one device buffer per GPU (`tlwedevices`).  level 0 = lvl0param,
    level 1 = lvl1param."""

    def __init__(self, level=0):
        self.level = int(level)
        # n + 1 / k N + 1 words of the parameter set the per-gate API runs on ("param_set"; the BASELINE set unless chosen otherwise)
        self.tlwehost = np.zeros(lib.cufhe_amd_ctxt_words(self.level), dtype=np.uint32)
        h = ctypes.c_void_p()
        check(lib.cufhe_amd_ctxt_create(self.level, _ptr(self.tlwehost), ctypes.byref(h)))
        self._h = h
        _lib.live.add(self)

    @property
    def tlwedevices(self):
        return [lib.cufhe_amd_ctxt_device_ptr(self._h, d) for d in range(GetGPUNum())]

    def release(self):
        if self._h and not _lib.closed:
            lib.cufhe_amd_ctxt_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Trlwe(Ctxt):
    """struct cuFHETRLWElvl1, include/cufhe_gpu.cuh:124-134: `trlwehost` ((k+1) N words) + a device buffer per GPU."""

    def __init__(self):
        self.level = 2
        self.tlwehost = np.zeros(lib.cufhe_amd_ctxt_words(2), dtype=np.uint32)      # (k+1) N words of the active parameter set
        self.trlwehost = self.tlwehost
        h = ctypes.c_void_p()
        check(lib.cufhe_amd_ctxt_create(2, _ptr(self.tlwehost), ctypes.byref(h)))
        self._h = h
        _lib.live.add(self)


TL_BOOTSTRAP, TL_REFRESH, TL_SEIKS = 100, 101, 102


def _trlwe_op(op, copying, out, inp, st):
    check(lib.cufhe_amd_enqueue_trlwe_op(st.device_id(), st.st(), op, 1 if copying else 0, out._h, inp._h))


def GateBootstrappingTLWE2TRLWElvl01NTT(out, inp, st):     # src/cufhe_gates_gpu.cu:96-104
    _trlwe_op(TL_BOOTSTRAP, True, out, inp, st)


def gGateBootstrappingTLWE2TRLWElvl01NTT(out, inp, st):    # :86-94
    _trlwe_op(TL_BOOTSTRAP, False, out, inp, st)


def Refresh(out, inp, st):                                 # :115-124
    _trlwe_op(TL_REFRESH, True, out, inp, st)


def gRefresh(out, inp, st):                                # :106-113
    _trlwe_op(TL_REFRESH, False, out, inp, st)


def gSampleExtractAndKeySwitch(out, inp, st):              # :126-135 (uploads in.trlwehost, as the reference does)
    CtxtCopyH2D(inp, st)
    _trlwe_op(TL_SEIKS, False, out, inp, st)


def SampleExtractAndKeySwitch(out, inp, st):               # :137-146
    gSampleExtractAndKeySwitch(out, inp, st)
    CtxtCopyD2H(out, st)


def CtxtCopyH2D(c, st):
    check(lib.cufhe_amd_enqueue_copy(st.device_id(), st.st(), c._h, 1))


def CtxtCopyD2H(c, st):
    check(lib.cufhe_amd_enqueue_copy(st.device_id(), st.st(), c._h, 0))


def CopyOnHost(out, inp):
    out.tlwehost[:] = inp.tlwehost


def Flush(device=0):
    """Launch the recorded gates of `device` without waiting (no reference counterpart:
    the reference launches at call time)."""
    check(lib.cufhe_amd_flush(device))


def _gate(op, copying, out, ins, st):
    hs = [c._h for c in ins] + [None] * (3 - len(ins))
    check(lib.cufhe_amd_enqueue_gate(st.device_id(), st.st(), op, 1 if copying else 0, out._h, *hs))


def _make2(op, copying):
    def f(out, in0, in1, st):
        _gate(op, copying, out, [in0, in1], st)
    return f


def _make1(op, copying):
    def f(out, in0, st):
        _gate(op, copying, out, [in0], st)
    return f


def _make3(op, copying):
    def f(out, inc, in1, in0, st):               # Mux(out, inc, in1, in0, st)
        _gate(op, copying, out, [inc, in1, in0], st)
    return f


And, gAnd = _make2(AND, True), _make2(AND, False)
AndYN, gAndYN = _make2(ANDYN, True), _make2(ANDYN, False)
AndNY, gAndNY = _make2(ANDNY, True), _make2(ANDNY, False)
Or, gOr = _make2(OR, True), _make2(OR, False)
OrYN, gOrYN = _make2(ORYN, True), _make2(ORYN, False)
OrNY, gOrNY = _make2(ORNY, True), _make2(ORNY, False)
Nand, gNand = _make2(NAND, True), _make2(NAND, False)
Nor, gNor = _make2(NOR, True), _make2(NOR, False)
Xor, gXor = _make2(XOR, True), _make2(XOR, False)
Xnor, gXnor = _make2(XNOR, True), _make2(XNOR, False)
Not, gNot = _make1(NOT, True), _make1(NOT, False)
Copy, gCopy = _make1(COPY, True), _make1(COPY, False)
Mux, gMux = _make3(MUX, True), _make3(MUX, False)
NMux, gNMux = _make3(NMUX, True), _make3(NMUX, False)


# ---- native batched entry points (what bench.py and the parity tests drive) ----
def gate_batch(ops, level, out, in0, in1=None, in2=None, count=None, device=0, stream=None):
    """ops: one op code or an int array of `count` codes; operands are DeviceBuffers holding
    `count` contiguous ciphertexts."""
    # n + 1 / k N + 1 words of the set the gate entry points run on NOW ("param_set"): the library dispatches level 0 and 1 to that set
    words = lib.cufhe_amd_ctxt_words(level) if level in (0, 1) else 1     # a bad level is rejected by the library
    if count is None:
        count = out.words // words
    if np.isscalar(ops):
        ops_arr, stride = np.array([ops], dtype=np.int32), 0
    else:
        ops_arr, stride = np.ascontiguousarray(ops, dtype=np.int32), 1
        assert ops_arr.size >= count
    check(lib.cufhe_amd_gate_batch(device, stream, level, count, _ptr(ops_arr), stride, out.ptr, in0.ptr,
                                   in1.ptr if in1 is not None else None,
                                   in2.ptr if in2 is not None else None, words))


def blind_rotate_batch(tlwe0, acc, count, steps=-1, device=0, stream=None):
    check(lib.cufhe_amd_blind_rotate_batch(device, stream, count, tlwe0.ptr, acc.ptr, steps))


def bootstrap_batch(out, inp, count, device=0, stream=None):
    """Bootstrap (src/bootstrap_gpu.cu:782-788): refresh `count` lvl0 ciphertexts."""
    check(lib.cufhe_amd_bootstrap_batch(device, stream, count, out.ptr, inp.ptr))


def keyswitch_batch(tlwe1, tlwe0, count, device=0, stream=None):
    check(lib.cufhe_amd_keyswitch_batch(device, stream, count, tlwe1.ptr, tlwe0.ptr))


def sample_extract_keyswitch_batch(trlwe, tlwe0, count, device=0, stream=None):
    check(lib.cufhe_amd_sample_extract_keyswitch_batch(device, stream, count, trlwe.ptr, tlwe0.ptr))


def refresh_batch(trlwe_in, trlwe_out, count, device=0, stream=None):
    check(lib.cufhe_amd_refresh_batch(device, stream, count, trlwe_in.ptr, trlwe_out.ptr))


def trgsw_to_ntt_batch(trgsw, trgsw_ntt, count, device=0, stream=None):
    check(lib.cufhe_amd_trgsw_to_ntt_batch(device, stream, count, trgsw.ptr, trgsw_ntt.ptr))


def cmux_batch(trgsw_ntt, c1, c0, res, count, device=0, stream=None):
    check(lib.cufhe_amd_cmux_batch(device, stream, count, trgsw_ntt.ptr, c1.ptr, c0.ptr, res.ptr))


def polymul_batch(a, b, res, count, device=0, stream=None):
    check(lib.cufhe_amd_polymul_batch(device, stream, count, a.ptr, b.ptr, res.ptr))


# ---- N = 2048 ring / 64-bit torus (configs[4]); gates on lvl0 ciphertexts ----
def lvl2_params():
    p = Lvl2Params()
    check(lib.cufhe_amd_lvl2_get_params(ctypes.byref(p)))
    return p


def lvl2_initialize(bk, ksk):
    """bk: the lvl02 bootstrapping key as uint64 torus words, ksk: the lvl20 key-switching key (uint32)."""
    bk = np.ascontiguousarray(bk, dtype=np.uint64).ravel()
    ksk = np.ascontiguousarray(ksk, dtype=np.uint32).ravel()
    check(lib.cufhe_amd_lvl2_initialize(_ptr(bk), bk.size, _ptr(ksk), ksk.size))


def lvl2_gate_batch(ops, out, in0, in1=None, in2=None, count=None, device=0, stream=None):
    words = LVL_WORDS[0]
    if count is None:
        count = out.words // words
    if np.isscalar(ops):
        ops_arr, stride = np.array([ops], dtype=np.int32), 0
    else:
        ops_arr, stride = np.ascontiguousarray(ops, dtype=np.int32), 1
        assert ops_arr.size >= count
    check(lib.cufhe_amd_lvl2_gate_batch(device, stream, count, _ptr(ops_arr), stride, out.ptr, in0.ptr,
                                        in1.ptr if in1 is not None else None,
                                        in2.ptr if in2 is not None else None, words))


def lvl2_blind_rotate_batch(tlwe0, acc, count, steps=-1, device=0, stream=None):
    check(lib.cufhe_amd_lvl2_blind_rotate_batch(device, stream, count, tlwe0.ptr, acc.ptr, steps))


def lvl2_keyswitch_batch(tlwe2, tlwe0, count, device=0, stream=None):
    check(lib.cufhe_amd_lvl2_keyswitch_batch(device, stream, count, tlwe2.ptr, tlwe0.ptr))


def polymul512_batch(a, b, res, count, device=0, stream=None):
    check(lib.cufhe_amd_polymul512_batch(device, stream, count, a.ptr, b.ptr, res.ptr))


def set_option(key, value):
    check(lib.cufhe_amd_set_option(key.encode(), int(value)))


def profile_enable(on=True, device=0):
    check(lib.cufhe_amd_profile_enable(device, 1 if on else 0))


def probe_clock(device=0):
    """Shader clock in Hz under an FP64 load, measured now (cufhe_amd_probe_clock)."""
    hz = ctypes.c_double(0.0)
    check(lib.cufhe_amd_probe_clock(device, ctypes.byref(hz)))
    return hz.value


def profile_get(device=0, reset=True):
    p = Profile()
    check(lib.cufhe_amd_profile_get(device, ctypes.byref(p), 1 if reset else 0))
    return p


def sched_stats(device=0, reset=False):
    """Counters of the per-gate API's scheduler: levels, launch sequences, copies (include/cufhe_amd.h)."""
    s = SchedStats()
    check(lib.cufhe_amd_sched_get_stats(device, ctypes.byref(s), 1 if reset else 0))
    return s


# ---- other parameter sets (include/cufhe_amd.h: cufhe_amd_ps_*) ----
def sched_trace(device=0, clear=True, max_entries=64):
    """the per-flush timeline of the per-gate API (cufhe_amd_sched_get_trace): list of dicts, oldest first"""
    buf = (_lib.GroupTrace * max_entries)()
    n = check(lib.cufhe_amd_sched_get_trace(int(device), buf, max_entries, int(bool(clear))))
    return [{f: getattr(buf[i], f) for f, _ in _lib.GroupTrace._fields_ if not f.startswith("pad")} for i in range(n)]


def ps_count():
    return lib.cufhe_amd_ps_count()


def ps_params(ps):
    p = PsParams()
    check(lib.cufhe_amd_ps_get_params(int(ps), ctypes.byref(p)))
    return p


def ps_index(name):
    for i in range(ps_count()):
        if ps_params(i).name.decode() == name:
            return i
    raise KeyError(name)


def ps_initialize(ps, bk, ksk):
    bk = np.ascontiguousarray(bk, dtype=np.uint32).ravel()
    ksk = np.ascontiguousarray(ksk, dtype=np.uint32).ravel()
    check(lib.cufhe_amd_ps_initialize(int(ps), _ptr(bk), bk.size, _ptr(ksk), ksk.size))


def ps_gate_batch(ps, ops, out, in0, in1=None, in2=None, count=None, device=0, stream=None, level=0):
    """level 0: blind rotate then key switch on n + 1 words; level 1: key switch then blind rotate on k N + 1 words"""
    words = ps_params(ps).lvl1_words if level else ps_params(ps).lvl0_words
    if count is None:
        count = out.words // words
    if np.isscalar(ops):
        ops_arr, stride = np.array([ops], dtype=np.int32), 0
    else:
        ops_arr, stride = np.ascontiguousarray(ops, dtype=np.int32), 1
        assert ops_arr.size >= count
    check(lib.cufhe_amd_ps_gate_batch_level(int(ps), device, stream, int(level), count, _ptr(ops_arr), stride, out.ptr, in0.ptr,
                                            in1.ptr if in1 is not None else None,
                                            in2.ptr if in2 is not None else None, words))


def ps_blind_rotate_batch(ps, tlwe0, acc, count, steps=-1, device=0, stream=None):
    check(lib.cufhe_amd_ps_blind_rotate_batch(int(ps), device, stream, count, tlwe0.ptr, acc.ptr, steps))


def ps_keyswitch_batch(ps, tlwe1, tlwe0, count, device=0, stream=None):
    check(lib.cufhe_amd_ps_keyswitch_batch(int(ps), device, stream, count, tlwe1.ptr, tlwe0.ptr))


def ps_trlwe_op_batch(ps, op, out, inp, count, device=0, stream=None):
    """op: TL_BOOTSTRAP (lvl0 TLWEs -> TRLWEs), TL_REFRESH (TRLWEs -> TRLWEs) or TL_SEIKS (TRLWEs -> lvl0 TLWEs), device buffers of the set's sizes"""
    check(lib.cufhe_amd_ps_trlwe_op_batch(int(ps), device, stream, int(op), count, out.ptr, inp.ptr))


def ps_trgsw_to_ntt_batch(ps, trgsw, trgsw_ntt, count, device=0, stream=None):
    """TRGSW2NTT on a set: trgsw[count][(k+1)l][k+1][N] torus words -> trgsw_ntt[count][limbs][(k+1)l][k+1][N] doubles (2 words each)"""
    check(lib.cufhe_amd_ps_trgsw_to_ntt_batch(int(ps), device, stream, count, trgsw.ptr, trgsw_ntt.ptr))


def ps_cmux_batch(ps, trgsw_ntt, c1, c0, res, count, device=0, stream=None):
    """CMUXNTT on a set: res = c0 + trgsw [x] (c1 - c0) on TRLWEs of (k+1) N words"""
    check(lib.cufhe_amd_ps_cmux_batch(int(ps), device, stream, count, trgsw_ntt.ptr, c1.ptr, c0.ptr, res.ptr))
