"""ctypes binding of the CPU oracle (oracle/libta_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SAN = os.environ.get("TA_SANITIZED") == "1"        # scripts/run_sanitized.sh: the ASan + UBSan builds (python needs LD_PRELOAD=libasan)
_SO = os.path.join(_ROOT, "oracle", "libta_oracle_san.so" if _SAN else "libta_oracle.so")

NONE = 0xFFFFFFFF
ALL, BEST = 0, 1
EDIT_NAMES = ["Match", "Mismatch", "AGap", "BGap", "Transpose"]


class Costs(C.Structure):
    _fields_ = [("mismatch_cost", C.c_uint8), ("gap_cost", C.c_uint8), ("start_gap_cost", C.c_uint8),
                ("has_transpose", C.c_uint8), ("transpose_cost", C.c_uint8)]


class Match(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("k", C.c_uint32), ("pad_", C.c_uint32)]


class Edit(C.Structure):
    _fields_ = [("edit", C.c_uint32), ("pad_", C.c_uint32), ("count", C.c_uint64)]


def build(force=False):
    src = os.path.join(_ROOT, "oracle", "ta_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-s"] + (["SAN=1"] if _SAN else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, sz, u32 = C.c_char_p, C.c_size_t, C.c_uint32
        cp = C.POINTER(Costs)
        epp, szp = C.POINTER(C.POINTER(Edit)), C.POINTER(C.c_size_t)
        mpp = C.POINTER(C.POINTER(Match))
        L.tao_costs_valid.argtypes = [cp]; L.tao_costs_valid.restype = C.c_int
        L.tao_costs_valid_search.argtypes = [cp]; L.tao_costs_valid_search.restype = C.c_int
        L.tao_hamming_naive.argtypes = [u8p, sz, u8p, sz]; L.tao_hamming_naive.restype = u32
        L.tao_hamming_search_naive_with_opts.argtypes = [u8p, sz, u8p, sz, u32, C.c_int, mpp]
        L.tao_hamming_search_naive_with_opts.restype = sz
        L.tao_hamming_search_simd_with_opts.argtypes = [u8p, sz, u8p, sz, u32, C.c_int, mpp, szp]
        L.tao_hamming_search_simd_with_opts.restype = C.c_int
        L.tao_levenshtein_naive_with_opts.argtypes = [u8p, sz, u8p, sz, C.c_int, cp, epp, szp]
        L.tao_levenshtein_naive_with_opts.restype = u32
        for name in ("tao_levenshtein_naive_k_with_opts", "tao_levenshtein_simd_k_with_opts"):
            f = getattr(L, name)
            f.argtypes = [u8p, sz, u8p, sz, u32, C.c_int, cp, epp, szp]
            f.restype = u32
        L.tao_levenshtein_select.argtypes = [sz, sz, u32, cp] + [C.POINTER(u32)] * 4
        L.tao_levenshtein_select.restype = None
        for name in ("tao_levenshtein", "tao_rdamerau", "tao_levenshtein_exp", "tao_rdamerau_exp"):
            f = getattr(L, name)
            f.argtypes = [u8p, sz, u8p, sz]
            f.restype = u32
        L.tao_levenshtein_exp_with_opts.argtypes = [u8p, sz, u8p, sz, C.c_int, cp, epp, szp]
        L.tao_levenshtein_exp_with_opts.restype = u32
        L.tao_levenshtein_search_naive_with_opts.argtypes = [u8p, sz, u8p, sz, u32, C.c_int, cp, C.c_int, mpp, szp]
        L.tao_levenshtein_search_naive_with_opts.restype = C.c_int
        L.tao_default_search_k.argtypes = [sz]; L.tao_default_search_k.restype = u32
        L.tao_band_cells.argtypes = [sz, sz, u32, cp]; L.tao_band_cells.restype = C.c_uint64
        for name in ("tao_levenshtein_k_batch",):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz, u32, cp, C.c_void_p, C.c_int]
            getattr(L, name).restype = None
        L.tao_levenshtein_k_batch_antidiag.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz, u32, cp, C.c_void_p,
                                                       C.c_int]
        L.tao_levenshtein_k_batch_antidiag.restype = C.c_int
        L.tao_levenshtein_k_batch_ladder.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz, u32, cp, C.c_void_p,
                                                     C.c_int, C.c_void_p]
        L.tao_levenshtein_k_batch_ladder.restype = C.c_int
        L.tao_have_avx2.argtypes = []; L.tao_have_avx2.restype = C.c_int
        L.tao_levenshtein_exp_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz, cp, C.c_void_p, C.c_int]
        L.tao_levenshtein_exp_batch.restype = None
        L.tao_hamming_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, sz, C.c_void_p, C.c_int]
        L.tao_hamming_batch.restype = None
        L.tao_max_threads.argtypes = []; L.tao_max_threads.restype = C.c_int
        L.tao_free.argtypes = [C.c_void_p]; L.tao_free.restype = None
        _lib = L
    return _lib


LEVENSHTEIN_COSTS = (1, 1, 0, None)
RDAMERAU_COSTS = (1, 1, 0, 1)


def mk_costs(c):
    """c = (mismatch, gap, start_gap, transpose|None) or a dict from kats.json."""
    if isinstance(c, dict):
        c = (c["mismatch"], c["gap"], c["start_gap"], c["transpose"])
    return Costs(c[0], c[1], c[2], 0 if c[3] is None else 1, 0 if c[3] is None else c[3])


def costs_valid(c):
    return bool(lib().tao_costs_valid(C.byref(mk_costs(c))))


def costs_valid_search(c):
    return bool(lib().tao_costs_valid_search(C.byref(mk_costs(c))))


def _take_edits(ep, n):
    out = [(EDIT_NAMES[ep[i].edit], int(ep[i].count)) for i in range(n.value)]
    if ep:
        lib().tao_free(ep)
    return out


def _take_matches(mp, n):
    out = [(int(mp[i].start), int(mp[i].end), int(mp[i].k)) for i in range(n)]
    if mp:
        lib().tao_free(mp)
    return out


def _opt(v):
    return None if v == NONE else int(v)


def hamming_naive(a, b):
    """None stands for the Rust panic (length mismatch)."""
    return _opt(lib().tao_hamming_naive(a, len(a), b, len(b)))


def hamming_search_naive_with_opts(needle, haystack, k, search_type):
    mp = C.POINTER(Match)()
    n = lib().tao_hamming_search_naive_with_opts(needle, len(needle), haystack, len(haystack), k, search_type, C.byref(mp))
    return _take_matches(mp, n)


def hamming_search_simd_with_opts(needle, haystack, k, search_type):
    """Raises ValueError where Rust panics (NUL byte in haystack)."""
    mp = C.POINTER(Match)()
    n = C.c_size_t()
    rc = lib().tao_hamming_search_simd_with_opts(needle, len(needle), haystack, len(haystack), k, search_type,
                                                 C.byref(mp), C.byref(n))
    if rc:
        raise ValueError("No zero/null bytes allowed in the string!")
    return _take_matches(mp, n.value)


def _lev(fn, a, b, k, trace_on, costs):
    ep = C.POINTER(Edit)()
    n = C.c_size_t()
    cs = mk_costs(costs)
    if k is None:
        d = fn(a, len(a), b, len(b), int(trace_on), C.byref(cs), C.byref(ep), C.byref(n))
    else:
        d = fn(a, len(a), b, len(b), k, int(trace_on), C.byref(cs), C.byref(ep), C.byref(n))
    tr = _take_edits(ep, n) if trace_on and d != NONE else None
    return _opt(d), tr


def levenshtein_naive_with_opts(a, b, trace_on=False, costs=LEVENSHTEIN_COSTS):
    return _lev(lib().tao_levenshtein_naive_with_opts, a, b, None, trace_on, costs)


def levenshtein_naive_k_with_opts(a, b, k, trace_on=False, costs=LEVENSHTEIN_COSTS):
    return _lev(lib().tao_levenshtein_naive_k_with_opts, a, b, k, trace_on, costs)


def levenshtein_simd_k_with_opts(a, b, k, trace_on=False, costs=LEVENSHTEIN_COSTS):
    return _lev(lib().tao_levenshtein_simd_k_with_opts, a, b, k, trace_on, costs)


def levenshtein_exp_with_opts(a, b, trace_on=False, costs=LEVENSHTEIN_COSTS):
    return _lev(lib().tao_levenshtein_exp_with_opts, a, b, None, trace_on, costs)


def levenshtein(a, b):
    return int(lib().tao_levenshtein(a, len(a), b, len(b)))


def rdamerau(a, b):
    return int(lib().tao_rdamerau(a, len(a), b, len(b)))


def levenshtein_exp(a, b):
    return int(lib().tao_levenshtein_exp(a, len(a), b, len(b)))


def rdamerau_exp(a, b):
    return int(lib().tao_rdamerau_exp(a, len(a), b, len(b)))


def levenshtein_select(a_len, b_len, k, costs=LEVENSHTEIN_COSTS):
    """-> (max_k, unit_k, cell_bits, lanes) of the reference's AVX2 dispatch ladder."""
    o = [C.c_uint32() for _ in range(4)]
    cs = mk_costs(costs)
    lib().tao_levenshtein_select(a_len, b_len, k, C.byref(cs), *[C.byref(x) for x in o])
    return tuple(int(x.value) for x in o)


def default_search_k(n):
    return int(lib().tao_default_search_k(n))


def band_cells(a_len, b_len, k, costs=LEVENSHTEIN_COSTS):
    cs = mk_costs(costs)
    return int(lib().tao_band_cells(a_len, b_len, k, C.byref(cs)))


def levenshtein_search_naive_with_opts(needle, haystack, k, search_type, costs=LEVENSHTEIN_COSTS, anchored=False):
    """Raises ValueError where Rust panics (check_search)."""
    mp = C.POINTER(Match)()
    n = C.c_size_t()
    cs = mk_costs(costs)
    rc = lib().tao_levenshtein_search_naive_with_opts(needle, len(needle), haystack, len(haystack), k, search_type,
                                                      C.byref(cs), int(anchored), C.byref(mp), C.byref(n))
    if rc:
        raise ValueError("invalid costs for search")
    return _take_matches(mp, n.value)


# ---------------------------------------------------------------- batch drivers (numpy CSR in, numpy out)
def _np():
    import numpy as np
    return np


def csr_from_list(strings):
    np = _np()
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in strings])
    blob = np.zeros(int(off[-1]) + 16, dtype=np.uint8)
    if off[-1]:
        blob[:int(off[-1])] = np.frombuffer(b"".join(strings), dtype=np.uint8)
    return blob, off


def csr_from_fixed(arr2d):
    np = _np()
    n, length = arr2d.shape
    blob = np.zeros(n * length + 16, dtype=np.uint8)
    blob[: n * length] = np.ascontiguousarray(arr2d).reshape(-1)
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(length))
    return blob, off


def max_threads():
    return int(lib().tao_max_threads())


def levenshtein_k_batch(a_csr, b_csr, k, costs=LEVENSHTEIN_COSTS, threads=0):
    """-> uint32 array, 0xFFFFFFFF == None"""
    np = _np()
    (ab, ao), (bb, bo) = a_csr, b_csr
    n = len(ao) - 1
    out = np.empty(n, dtype=np.uint32)
    cs = mk_costs(costs)
    lib().tao_levenshtein_k_batch(ab.ctypes.data, ao.ctypes.data, bb.ctypes.data, bo.ctypes.data, n, k, C.byref(cs),
                                  out.ctypes.data, threads or max_threads())
    return out


def levenshtein_k_batch_antidiag(a_csr, b_csr, k, costs=LEVENSHTEIN_COSTS, threads=0):
    """The anti-diagonal auto-vectorised restatement (oracle/ta_oracle_simd.c); None when it does not apply (affine gaps)."""
    np = _np()
    (ab, ao), (bb, bo) = a_csr, b_csr
    n = len(ao) - 1
    out = np.empty(n, dtype=np.uint32)
    cs = mk_costs(costs)
    rc = lib().tao_levenshtein_k_batch_antidiag(ab.ctypes.data, ao.ctypes.data, bb.ctypes.data, bo.ctypes.data, n, k,
                                                C.byref(cs), out.ctypes.data, threads or max_threads())
    return out if rc == 0 else None


def have_avx2():
    return bool(lib().tao_have_avx2())


def levenshtein_k_batch_ladder(a_csr, b_csr, k, costs=LEVENSHTEIN_COSTS, threads=0, hist=False):
    """The hand-written AVX2 u8 anti-diagonal restatement with the width ladder 8 -> 16 -> 32 around it
    (oracle/ta_oracle_avx2.c); None when it does not apply (no AVX2 on the host, affine gaps).  hist=True also returns the
    six class counters (no DP / 32 / 64 / 128 / 256 u8 lanes / wider cells)."""
    np = _np()
    (ab, ao), (bb, bo) = a_csr, b_csr
    n = len(ao) - 1
    out = np.empty(n, dtype=np.uint32)
    h = np.zeros(6, dtype=np.uint64)
    cs = mk_costs(costs)
    rc = lib().tao_levenshtein_k_batch_ladder(ab.ctypes.data, ao.ctypes.data, bb.ctypes.data, bo.ctypes.data, n, k,
                                              C.byref(cs), out.ctypes.data, threads or max_threads(), h.ctypes.data)
    if rc != 0:
        return (None, None) if hist else None
    return (out, [int(x) for x in h]) if hist else out


def levenshtein_exp_batch(a_csr, b_csr, costs=LEVENSHTEIN_COSTS, threads=0):
    np = _np()
    (ab, ao), (bb, bo) = a_csr, b_csr
    n = len(ao) - 1
    out = np.empty(n, dtype=np.uint32)
    cs = mk_costs(costs)
    lib().tao_levenshtein_exp_batch(ab.ctypes.data, ao.ctypes.data, bb.ctypes.data, bo.ctypes.data, n, C.byref(cs),
                                    out.ctypes.data, threads or max_threads())
    return out


def hamming_batch(a_csr, b_csr, threads=0):
    np = _np()
    (ab, ao), (bb, bo) = a_csr, b_csr
    n = len(ao) - 1
    out = np.empty(n, dtype=np.uint32)
    lib().tao_hamming_batch(ab.ctypes.data, ao.ctypes.data, bb.ctypes.data, bo.ctypes.data, n, out.ctypes.data,
                            threads or max_threads())
    return out
