"""Loads libtriple_accel_amd.so (the C ABI of include/triple_accel_amd.h) through ctypes.

There is no pure-Python or CPU fallback: if the library is missing this raises, and if no
HIP device is present every compute call raises TripleAccelError(TA_ERR_HIP).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtriple_accel_amd.so")

NONE = 0xFFFFFFFF
TA_OK, TA_ERR_LEN_MISMATCH, TA_ERR_NULL_BYTE, TA_ERR_BAD_COSTS, TA_ERR_HIP, TA_ERR_ARG, TA_ERR_UNSUPPORTED, \
    TA_ERR_CAPACITY, TA_ERR_DIV_ZERO = range(9)

# every symbol include/triple_accel_amd.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "ta_levenshtein_costs", "ta_rdamerau_costs", "ta_edit_costs_new", "ta_edit_costs_check_search",
    "ta_version", "ta_status_str", "ta_device_count", "ta_last_error", "ta_last_kernel_name", "ta_set_option", "ta_queue_create", "ta_queue_push", "ta_queue_flush", "ta_queue_destroy", "ta_levenshtein_k_batch_alphabet", "ta_levenshtein_search_first", "ta_levenshtein_search_first_dev", "ta_levenshtein_select",
    "ta_last_launch_info", "ta_hamming", "ta_levenshtein_simd_k_with_opts", "ta_levenshtein_trace",
    "ta_levenshtein_exp_trace", "ta_levenshtein_simd_k",
    "ta_levenshtein", "ta_rdamerau", "ta_levenshtein_exp", "ta_levenshtein_exp_with_opts", "ta_rdamerau_exp",
    "ta_levenshtein_search_simd_with_opts", "ta_levenshtein_search", "ta_hamming_search_simd_with_opts",
    "ta_hamming_search", "ta_hamming_search_naive_with_opts", "ta_free", "ta_thread_release", "ta_levenshtein_k_batch", "ta_levenshtein_exp_batch", "ta_hamming_batch",
    "ta_levenshtein_search_dev", "ta_hamming_search_dev", "ta_search_fold_best", "ta_search_best_hits_dev",
    "ta_levenshtein_search_best_dev", "ta_levenshtein_trace_batch", "ta_hamming_search_dev_sorted", "ta_levenshtein_search_resume",
    # the device set (ta_multi.hip)
    "ta_levenshtein_trace_batch_packed",
    "ta_set_devices", "ta_get_devices", "ta_levenshtein_k_batch_host", "ta_levenshtein_exp_batch_host", "ta_hamming_batch_host", "ta_levenshtein_trace_batch_host",
    "ta_sharded_pairs_upload", "ta_sharded_pairs_levenshtein_k", "ta_sharded_pairs_levenshtein_exp", "ta_sharded_pairs_hamming",
    "ta_sharded_pairs_time_levenshtein_k", "ta_sharded_pairs_shards", "ta_sharded_pairs_free",
    "ta_sharded_haystack_upload", "ta_sharded_haystack_levenshtein_search", "ta_sharded_haystack_hamming_search",
    "ta_sharded_haystack_shards", "ta_sharded_haystack_free",
]


class EditCostsC(C.Structure):
    _fields_ = [("mismatch_cost", C.c_uint8), ("gap_cost", C.c_uint8), ("start_gap_cost", C.c_uint8),
                ("has_transpose", C.c_uint8), ("transpose_cost", C.c_uint8)]


class MatchC(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("k", C.c_uint32), ("pad_", C.c_uint32)]


class EditC(C.Structure):
    _fields_ = [("edit", C.c_uint32), ("pad_", C.c_uint32), ("count", C.c_uint64)]


class LevSelectC(C.Structure):
    _fields_ = [("max_k", C.c_uint32), ("unit_k", C.c_uint32), ("cell_bits", C.c_uint32), ("ref_lanes", C.c_uint32)]


class LaunchInfoC(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("kernel", "diags_per_lane", "lanes_per_pair", "pairs_per_wave",
                                          "band_offset", "cell_bits", "affine", "transpose", "grid", "lds_bytes")]


class StringsC(C.Structure):
    _fields_ = [("blob", C.c_void_p), ("off", C.c_void_p), ("stride", C.c_uint64), ("len", C.c_uint64),
                ("max_len", C.c_uint64)]


class TripleAccelError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().ta_status_str(status).decode()
        if detail:
            msg += ": " + detail
        super().__init__("triple_accel_amd: " + msg)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("triple_accel_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` or `make -C triple_accel_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    # PyTorch (device memory, streams) ships its own HIP runtime: let it bring the GPU up before this library makes its first
    # HIP call -- the other way round torch reports "No HIP GPUs are available" on this stack.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    u8p, sz, u32, i32 = C.c_char_p, C.c_size_t, C.c_uint32, C.c_int
    cp, u32p = C.POINTER(EditCostsC), C.POINTER(C.c_uint32)
    mpp, szp = C.POINTER(C.POINTER(MatchC)), C.POINTER(C.c_size_t)
    sp = C.POINTER(StringsC)

    def sig(name, res, args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args

    sig("ta_version", C.c_char_p, [])
    sig("ta_status_str", C.c_char_p, [i32])
    sig("ta_device_count", i32, [])
    sig("ta_last_error", C.c_char_p, [])
    sig("ta_last_kernel_name", C.c_char_p, [])
    sig("ta_set_option", C.c_int, [C.c_int, C.c_int])
    sig("ta_queue_create", i32, [u32, cp, C.POINTER(C.c_void_p)])
    sig("ta_queue_push", i32, [C.c_void_p, u8p, sz, u8p, sz, C.POINTER(sz)])
    sig("ta_queue_flush", i32, [C.c_void_p, C.POINTER(C.POINTER(u32)), C.POINTER(sz)])
    sig("ta_queue_destroy", None, [C.c_void_p])
    sig("ta_levenshtein_costs", EditCostsC, [])
    sig("ta_rdamerau_costs", EditCostsC, [])
    sig("ta_edit_costs_new", i32, [C.c_uint8, C.c_uint8, C.c_uint8, i32, C.c_uint8, cp])
    sig("ta_edit_costs_check_search", i32, [cp])
    sig("ta_levenshtein_select", i32, [sz, sz, u32, cp, C.POINTER(LevSelectC)])
    sig("ta_last_launch_info", i32, [C.POINTER(LaunchInfoC)])
    sig("ta_hamming", i32, [u8p, sz, u8p, sz, u32p])
    sig("ta_levenshtein_simd_k_with_opts", i32, [u8p, sz, u8p, sz, u32, i32, cp, u32p])
    epp = C.POINTER(C.POINTER(EditC))
    sig("ta_levenshtein_trace", i32, [u8p, sz, u8p, sz, u32, cp, u32p, epp, szp])
    sig("ta_levenshtein_exp_trace", i32, [u8p, sz, u8p, sz, cp, u32p, epp, szp])
    sig("ta_levenshtein_simd_k", i32, [u8p, sz, u8p, sz, u32, u32p])
    for n in ("ta_levenshtein", "ta_rdamerau", "ta_levenshtein_exp", "ta_rdamerau_exp"):
        sig(n, i32, [u8p, sz, u8p, sz, u32p])
    sig("ta_levenshtein_exp_with_opts", i32, [u8p, sz, u8p, sz, i32, cp, u32p])
    sig("ta_levenshtein_search_simd_with_opts", i32, [u8p, sz, u8p, sz, u32, i32, cp, i32, mpp, szp])
    sig("ta_levenshtein_search", i32, [u8p, sz, u8p, sz, mpp, szp])
    sig("ta_levenshtein_search_resume", i32, [u8p, sz, u8p, sz, u32, cp, i32, mpp, szp])
    sig("ta_levenshtein_search_first", i32, [u8p, sz, u8p, sz, u32, cp, i32, C.POINTER(MatchC), C.POINTER(C.c_int)])
    sig("ta_levenshtein_search_first_dev", i32, [u8p, sz, C.c_void_p, sz, u32, cp, C.c_uint64, C.POINTER(MatchC), C.POINTER(C.c_int), C.c_void_p])
    sig("ta_hamming_search_simd_with_opts", i32, [u8p, sz, u8p, sz, u32, i32, mpp, szp])
    sig("ta_hamming_search", i32, [u8p, sz, u8p, sz, mpp, szp])
    sig("ta_hamming_search_naive_with_opts", i32, [u8p, sz, u8p, sz, u32, i32, mpp, szp])
    sig("ta_free", None, [C.c_void_p])
    sig("ta_thread_release", None, [])
    sig("ta_levenshtein_k_batch", i32, [sp, sp, sz, u32, cp, C.c_void_p, C.c_void_p])
    sig("ta_levenshtein_k_batch_alphabet", i32, [sp, sp, sz, u32, cp, u8p, sz, C.c_void_p, C.c_void_p])
    sig("ta_levenshtein_exp_batch", i32, [sp, sp, sz, cp, C.c_void_p, C.c_void_p])
    sig("ta_levenshtein_trace_batch", i32, [sp, sp, sz, u32, cp, C.c_void_p, C.c_void_p, C.c_void_p, sz, C.c_void_p])
    sig("ta_levenshtein_trace_batch_packed", i32, [sp, sp, sz, u32, cp, C.c_void_p, C.c_void_p, C.c_void_p, sz, C.c_void_p])
    sig("ta_hamming_batch", i32, [sp, sp, sz, C.c_void_p, C.c_void_p])
    sig("ta_levenshtein_search_dev", i32, [u8p, sz, C.c_void_p, sz, u32, cp, i32, C.c_uint64, C.c_uint64,
                                           C.c_void_p, sz, C.POINTER(C.c_uint64), C.c_void_p])
    sig("ta_hamming_search_dev", i32, [u8p, sz, C.c_void_p, sz, u32, C.c_uint64, C.c_void_p, sz,
                                       C.POINTER(C.c_uint64), C.c_void_p])
    sig("ta_hamming_search_dev_sorted", i32, [u8p, sz, C.c_void_p, sz, u32, C.c_uint64, C.c_void_p, sz, mpp, szp, C.c_void_p])
    sig("ta_search_fold_best", sz, [C.POINTER(MatchC), sz, u32, i32])
    sig("ta_levenshtein_search_best_dev", i32, [u8p, sz, C.c_void_p, sz, u32, cp, C.c_uint64, C.c_uint64,
                                                C.c_void_p, sz, C.POINTER(C.c_uint64), mpp, szp, C.c_void_p])
    sig("ta_search_best_hits_dev", i32, [C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(MatchC)), C.POINTER(sz), C.c_void_p])
    vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
    sig("ta_set_devices", i32, [C.POINTER(C.c_int), sz])
    sig("ta_get_devices", i32, [C.POINTER(C.c_int), sz, szp])
    sig("ta_levenshtein_k_batch_host", i32, [sp, sp, sz, u32, cp, vp])
    sig("ta_levenshtein_exp_batch_host", i32, [sp, sp, sz, cp, vp])
    sig("ta_hamming_batch_host", i32, [sp, sp, sz, vp])
    sig("ta_levenshtein_trace_batch_host", i32, [sp, sp, sz, u32, cp, vp, vp, vp, sz])
    sig("ta_sharded_pairs_upload", i32, [sp, sp, sz, sz, vpp])
    sig("ta_sharded_pairs_levenshtein_k", i32, [vp, u32, cp, vp])
    sig("ta_sharded_pairs_levenshtein_exp", i32, [vp, cp, vp])
    sig("ta_sharded_pairs_hamming", i32, [vp, vp])
    sig("ta_sharded_pairs_time_levenshtein_k", i32, [vp, u32, cp, i32, C.POINTER(C.c_float)])
    sig("ta_sharded_pairs_shards", i32, [vp, szp, szp])
    sig("ta_sharded_pairs_free", None, [vp])
    sig("ta_sharded_haystack_upload", i32, [vp, sz, sz, sz, vpp])
    sig("ta_sharded_haystack_levenshtein_search", i32, [vp, u8p, sz, u32, i32, cp, mpp, szp])
    sig("ta_sharded_haystack_hamming_search", i32, [vp, u8p, sz, u32, i32, mpp, szp])
    sig("ta_sharded_haystack_shards", i32, [vp, szp, szp])
    sig("ta_sharded_haystack_free", None, [vp])
    _lib = L
    return L


def check(status):
    if status != TA_OK:
        raise TripleAccelError(status, lib().ta_last_error().decode() if status in (TA_ERR_HIP, TA_ERR_ARG) else "")
