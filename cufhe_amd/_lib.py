"""ctypes binding of libcufhe_amd.so (the C ABI of include/cufhe_amd.h).

The library is the product; there is no CPU fallback.  Importing this module when the
shared object has not been built raises ImportError with the build command.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CUFHE_AMD_LIBRARY: load another build of the same library (the diagnostic builds of cufhe_amd/build.py)
LIB_PATH = os.environ.get("CUFHE_AMD_LIBRARY") or os.path.join(_HERE, "libcufhe_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()' at the repo root)")

lib = ctypes.CDLL(LIB_PATH)

c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_void = ctypes.c_void_p


class Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in
                ("n", "N", "nbit", "k", "l", "Bgbit", "t", "basebit", "mu", "lvl0_words", "lvl1_words")] + \
               [(n, ctypes.c_uint64) for n in ("bk_words", "ksk_words", "bk_ntt_bytes")]


class Lvl2Params(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in ("n", "N", "nbit", "k", "l", "Bgbit", "t", "basebit",
                                               "lvl0_words", "lvl2_words")] + \
               [(k, ctypes.c_uint64) for k in ("mu", "bk_words", "ksk_words", "bk_ntt_bytes")]


class Profile(ctypes.Structure):
    _fields_ = [("blind_rotate_ms", ctypes.c_double), ("blind_rotate_launches", ctypes.c_uint64),
                ("blind_rotations", ctypes.c_uint64), ("keyswitch_ms", ctypes.c_double),
                ("keyswitch_launches", ctypes.c_uint64), ("keyswitches", ctypes.c_uint64)]


class PsParams(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 32)] + \
               [(k, ctypes.c_uint32) for k in ("n", "N", "nbit", "k", "l", "Bgbit", "t", "basebit", "key_limbs",
                                               "key_limb_bits", "mu", "lvl0_words", "lvl1_words", "small_ntt_modulus")] + \
               [(k, ctypes.c_uint64) for k in ("bk_words", "ksk_words", "bk_ntt_bytes")]


class ParamNumbers(ctypes.Structure):
    """cufhe_amd_param_numbers: the numbers a caller was compiled with (include/cufhe_amd.h)"""
    _fields_ = [(k, ctypes.c_uint32) for k in ("n", "nbit", "k", "l", "Bgbit", "t", "basebit", "small_ntt_modulus")]


class SchedStats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint64) for k in ("gates", "groups", "levels", "launch_sequences", "uploads",
                                               "uploads_shared", "downloads", "forced_syncs", "max_level_gates",
                                               "cross_stream_waits", "record_ns", "retire_ns", "launch_ns", "renames",
                                               "worker_cpus", "home_copies", "two_lane_groups", "two_lane_launches")]


class GroupTrace(ctypes.Structure):
    _fields_ = [("id", ctypes.c_uint64), ("levels", ctypes.c_uint32), ("gates", ctypes.c_uint32), ("stream", ctypes.c_uint32),
                ("pad", ctypes.c_uint32), ("in_bytes", ctypes.c_uint64), ("out_bytes", ctypes.c_uint64)] + \
               [(k, ctypes.c_int64) for k in ("t_queued", "t_launch_begin", "t_gather_end", "t_submit_end", "t_done_seen", "t_delivered")] + \
               [(k, ctypes.c_float) for k in ("dev_h2d_ms", "dev_body_ms", "dev_d2h_ms", "pad2")]


# every symbol include/cufhe_amd.h declares, with its signature
SIGNATURES = {
    "cufhe_amd_get_params": (ctypes.c_int, [ctypes.POINTER(Params)]),
    "cufhe_amd_last_error": (ctypes.c_char_p, []),
    "cufhe_amd_set_gpu_num": (ctypes.c_int, [ctypes.c_int]),
    "cufhe_amd_get_gpu_num": (ctypes.c_int, []),
    "cufhe_amd_device_count": (ctypes.c_int, []),
    "cufhe_amd_device_identity": (ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]),
    "cufhe_amd_initialize_ntt": (ctypes.c_int, []),
    "cufhe_amd_initialize": (ctypes.c_int, [c_void, ctypes.c_size_t, c_void, ctypes.c_size_t]),
    "cufhe_amd_cleanup": (ctypes.c_int, []),
    "cufhe_amd_synchronize": (ctypes.c_int, []),
    "cufhe_amd_stream_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_void)]),
    "cufhe_amd_stream_destroy": (ctypes.c_int, [ctypes.c_int, c_void]),
    "cufhe_amd_stream_query": (ctypes.c_int, [ctypes.c_int, c_void]),
    "cufhe_amd_stream_synchronize": (ctypes.c_int, [ctypes.c_int, c_void]),
    "cufhe_amd_malloc": (ctypes.c_int, [ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(c_void)]),
    "cufhe_amd_free": (ctypes.c_int, [ctypes.c_int, c_void]),
    "cufhe_amd_host_register": (ctypes.c_int, [c_void, ctypes.c_size_t]),
    "cufhe_amd_host_unregister": (ctypes.c_int, [c_void]),
    "cufhe_amd_memcpy_h2d": (ctypes.c_int, [ctypes.c_int, c_void, c_void, c_void, ctypes.c_size_t]),
    "cufhe_amd_memcpy_d2h": (ctypes.c_int, [ctypes.c_int, c_void, c_void, c_void, ctypes.c_size_t]),
    "cufhe_amd_gate": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, c_void]),
    "cufhe_amd_gate_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_int, ctypes.c_size_t, c_void, ctypes.c_int,
                                            c_void, c_void, c_void, c_void, ctypes.c_size_t]),
    "cufhe_amd_gate_list": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_int, ctypes.c_size_t, c_void,
                                           c_void, c_void, c_void, c_void]),
    "cufhe_amd_ctxt_create": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.POINTER(c_void)]),
    "cufhe_amd_ctxt_destroy": (ctypes.c_int, [c_void]),
    "cufhe_amd_ctxt_device_ptr": (c_void, [c_void, ctypes.c_int]),
    "cufhe_amd_enqueue_gate": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, c_void]),
    "cufhe_amd_enqueue_trlwe_op": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_int, ctypes.c_int, c_void, c_void]),
    "cufhe_amd_enqueue_cmux": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_int, c_void, c_void, c_void, c_void]),
    "cufhe_amd_trgsw_to_ntt_host": (ctypes.c_int, [ctypes.c_int, c_void, c_void, c_void]),
    "cufhe_amd_trgsw_to_ntt": (ctypes.c_int, [ctypes.c_int, c_void, c_void, c_void]),
    "cufhe_amd_enqueue_copy": (ctypes.c_int, [ctypes.c_int, c_void, c_void, ctypes.c_int]),
    "cufhe_amd_flush": (ctypes.c_int, [ctypes.c_int]),
    "cufhe_amd_sched_stream_query": (ctypes.c_int, [ctypes.c_int, c_void]),
    "cufhe_amd_sched_get_stats": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(SchedStats), ctypes.c_int]),
    "cufhe_amd_sched_get_trace": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(GroupTrace), ctypes.c_int, ctypes.c_int]),
    "cufhe_amd_blind_rotate_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void, ctypes.c_int]),
    "cufhe_amd_keyswitch_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void]),
    "cufhe_amd_sample_extract_keyswitch_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void]),
    "cufhe_amd_refresh_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void]),
    "cufhe_amd_trgsw_to_ntt_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void]),
    "cufhe_amd_cmux_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void, c_void, c_void]),
    "cufhe_amd_polymul_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void, c_void]),
    "cufhe_amd_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_long]),
    "cufhe_amd_profile_enable": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "cufhe_amd_profile_get": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(Profile), ctypes.c_int]),
    "cufhe_amd_probe_clock": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "cufhe_amd_polymul512_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void, c_void]),
    "cufhe_amd_bootstrap_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void]),
    "cufhe_amd_device_cus": (ctypes.c_int, [ctypes.c_int]),
    "cufhe_amd_device_mem_info": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "cufhe_amd_stream_fence": (ctypes.c_int, [ctypes.c_int, c_void]),
    "cufhe_amd_find_param_set": (ctypes.c_int, [ctypes.POINTER(ParamNumbers)]),
    "cufhe_amd_initialize_params": (ctypes.c_int, [ctypes.POINTER(ParamNumbers), c_void, ctypes.c_size_t, c_void, ctypes.c_size_t]),
    "cufhe_amd_ps_trgsw_to_ntt_batch": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void]),
    "cufhe_amd_ps_cmux_batch": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void, c_void, c_void]),
    "cufhe_amd_ps_count": (ctypes.c_int, []),
    "cufhe_amd_ps_get_params": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(PsParams)]),
    "cufhe_amd_ps_initialize": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, ctypes.c_size_t]),
    "cufhe_amd_ps_gate_batch": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_void, ctypes.c_size_t, c_void, ctypes.c_int,
                                               c_void, c_void, c_void, c_void, ctypes.c_size_t]),
    "cufhe_amd_ps_gate_batch_level": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int, ctypes.c_size_t, c_void, ctypes.c_int,
                                                     c_void, c_void, c_void, c_void, ctypes.c_size_t]),
    "cufhe_amd_ctxt_words": (ctypes.c_int, [ctypes.c_int]),
    "cufhe_amd_ps_blind_rotate_batch": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void, ctypes.c_int]),
    "cufhe_amd_ps_keyswitch_batch": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void]),
    "cufhe_amd_ps_trlwe_op_batch": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_void, ctypes.c_int, ctypes.c_size_t, c_void, c_void]),
    "cufhe_amd_lvl2_get_params": (ctypes.c_int, [ctypes.POINTER(Lvl2Params)]),
    "cufhe_amd_lvl2_initialize": (ctypes.c_int, [c_void, ctypes.c_size_t, c_void, ctypes.c_size_t]),
    "cufhe_amd_lvl2_gate_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, ctypes.c_int,
                                                 c_void, c_void, c_void, c_void, ctypes.c_size_t]),
    "cufhe_amd_lvl2_blind_rotate_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void, ctypes.c_int]),
    "cufhe_amd_lvl2_keyswitch_batch": (ctypes.c_int, [ctypes.c_int, c_void, ctypes.c_size_t, c_void, c_void]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = the .so is stale: rebuild
    _fn.restype = _res
    _fn.argtypes = _args


class CufheAmdError(RuntimeError):
    pass


def check(rc):
    """Raise on a negative status, mirroring the reference's abort-on-error
    (include/details/error_gpu.cuh:40-60) as an exception."""
    if rc < 0:
        raise CufheAmdError(f"cufhe_amd error {rc}: {lib.cufhe_amd_last_error().decode()}")
    return rc


# Deterministic teardown: at interpreter exit, while Python and the HIP runtime are both still fully
# alive, release every live Ctxt / DeviceBuffer (unpinning the host memory the ciphertexts
# registered) and the library's device state; __del__ hooks that run later are no-ops.
import atexit  # noqa: E402
import weakref  # noqa: E402

closed = False
live = weakref.WeakSet()        # objects with a release() method (api.Ctxt, api.DeviceBuffer)


def _shutdown():
    global closed
    if closed:
        return
    for obj in list(live):
        try:
            obj.release()
        except Exception:
            pass
    closed = True
    try:
        lib.cufhe_amd_cleanup()
    except Exception:
        pass


atexit.register(_shutdown)
