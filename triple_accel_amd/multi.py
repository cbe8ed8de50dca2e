"""The device set: the multi-GPU split behind the C ABI (include/triple_accel_amd.h, "the device set"; csrc/ta_multi.hip).

One process, host memory in, host memory out -- the reference's calling convention (src/levenshtein.rs:714-720, 1911-1918, 2508-2511;
src/hamming.rs:454-475) -- with the pairs / the haystack partitioned over the GPUs of the node inside the library.  `triple_accel_amd.dist`
is the other form (one process per GPU, torch.distributed); both produce the one-device results bit for bit.

No torch here: the strings are numpy arrays / bytes in host memory, the answers numpy arrays.
"""
import ctypes as _C

import numpy as np

from . import _native as _n
from . import LEVENSHTEIN_COSTS, Match, SearchType, _costs, _raise


def set_devices(devices=None):
    """The GPUs the host entry points fan out over (None: every visible one, the default).  An id may be listed more than once: that many
    workers share the device (how a one-GPU box runs the N-way logic)."""
    if not devices:
        _raise(_n.lib().ta_set_devices(None, 0))
        return
    arr = (_C.c_int * len(devices))(*[int(d) for d in devices])
    _raise(_n.lib().ta_set_devices(arr, len(devices)))


def get_devices():
    n = _C.c_size_t(0)
    _raise(_n.lib().ta_get_devices(None, 0, _C.byref(n)))
    arr = (_C.c_int * max(n.value, 1))()
    _raise(_n.lib().ta_get_devices(arr, n.value, _C.byref(n)))
    return [int(arr[i]) for i in range(n.value)]


class HostStrings:
    """One side of a batch in HOST memory: a list of bytes (-> CSR), or an (n, len) uint8 array (strided, no copy if contiguous)."""

    def __init__(self, strings):
        if isinstance(strings, np.ndarray) and strings.ndim == 2:
            arr = np.ascontiguousarray(strings, dtype=np.uint8)
            self.n, length = arr.shape
            self.blob, self.off = arr.reshape(-1), None
            self.c = _n.StringsC(self.blob.ctypes.data if self.blob.size else 0, 0, length, length, length)
        else:
            strings = list(strings)
            self.n = len(strings)
            lens = np.fromiter((len(s) for s in strings), dtype=np.uint64, count=self.n)
            self.off = np.zeros(self.n + 1, dtype=np.uint64)
            np.cumsum(lens, out=self.off[1:])
            self.blob = np.frombuffer(b"".join(bytes(s) for s in strings), dtype=np.uint8).copy() if self.n and int(self.off[-1]) else np.zeros(1, dtype=np.uint8)
            self.c = _n.StringsC(self.blob.ctypes.data, self.off.ctypes.data, 0, 0, 0)

    @classmethod
    def of(cls, x):
        return x if isinstance(x, cls) else cls(x)


def _pairs(a, b):
    a, b = HostStrings.of(a), HostStrings.of(b)
    if a.n != b.n:
        raise ValueError("both sides of a batch hold the same number of strings")
    return a, b


def levenshtein_k_batch_host(a, b, k, costs=LEVENSHTEIN_COSTS):
    """[levenshtein_simd_k_with_opts(a_i, b_i, k, False, costs)] over the device set -> uint32 array (0xFFFFFFFF = None)."""
    a, b = _pairs(a, b)
    c = _costs(costs)._c()
    out = np.empty(a.n, dtype=np.uint32)
    _raise(_n.lib().ta_levenshtein_k_batch_host(_C.byref(a.c), _C.byref(b.c), a.n, int(k) & 0xFFFFFFFF, _C.byref(c), out.ctypes.data))
    return out


def levenshtein_exp_batch_host(a, b, costs=LEVENSHTEIN_COSTS):
    a, b = _pairs(a, b)
    c = _costs(costs)._c()
    out = np.empty(a.n, dtype=np.uint32)
    _raise(_n.lib().ta_levenshtein_exp_batch_host(_C.byref(a.c), _C.byref(b.c), a.n, _C.byref(c), out.ctypes.data))
    return out


def hamming_batch_host(a, b):
    a, b = _pairs(a, b)
    out = np.empty(a.n, dtype=np.uint32)
    _raise(_n.lib().ta_hamming_batch_host(_C.byref(a.c), _C.byref(b.c), a.n, out.ctypes.data))
    return out


_EDIT_NAMES = ("Match", "Mismatch", "AGap", "BGap", "Transpose")


def levenshtein_trace_batch_host(a, b, k, costs=LEVENSHTEIN_COSTS, cap=None, as_lists=True):
    """[levenshtein_simd_k_with_opts(a_i, b_i, k, True, costs)] over the device set: -> (distances uint32 (0xFFFFFFFF = None), scripts) with
    scripts[i] = [(edit name, count), ...] front to back ([] for None); as_lists=False: the raw (n, cap) packed words and the run counts
    instead (word = (edit type << 29) | count, a script in the LAST min(n_edits, cap) words of its row)."""
    a, b = _pairs(a, b)
    c = _costs(costs)._c()
    if cap is None:
        longest = 1
        for side in (a, b):
            longest = max(longest, int(np.diff(side.off).max()) if side.off is not None and side.n else int(side.c.len))
        cap = int(min(2 * (int(k) & 0xFFFFFFFF) + 1, 2 * longest + 2))
    out = np.empty(a.n, dtype=np.uint32)
    packed = np.zeros((a.n, cap), dtype=np.uint32)
    n_edits = np.zeros(a.n, dtype=np.uint32)
    _raise(_n.lib().ta_levenshtein_trace_batch_host(_C.byref(a.c), _C.byref(b.c), a.n, int(k) & 0xFFFFFFFF, _C.byref(c), out.ctypes.data,
                                                    packed.ctypes.data, n_edits.ctypes.data, cap))
    if not as_lists:
        return out, packed, n_edits
    scripts = []
    for i in range(a.n):
        have = min(int(n_edits[i]), cap)
        if int(n_edits[i]) > cap:
            raise ValueError("levenshtein_trace_batch_host: pair %d has a script of %d runs, cap = %d" % (i, int(n_edits[i]), cap))
        scripts.append([(_EDIT_NAMES[int(w) >> 29], int(w) & 0x1FFFFFFF) for w in packed[i, cap - have:]])
    return out, scripts


class ShardedPairs:
    """A pair batch uploaded once and kept resident, sharded over the device set."""

    def __init__(self, a, b, n_shards=0):
        a, b = _pairs(a, b)
        self.n = a.n
        self._h = _C.c_void_p()
        _raise(_n.lib().ta_sharded_pairs_upload(_C.byref(a.c), _C.byref(b.c), a.n, int(n_shards), _C.byref(self._h)))

    @property
    def n_shards(self):
        s, n = _C.c_size_t(0), _C.c_size_t(0)
        _raise(_n.lib().ta_sharded_pairs_shards(self._h, _C.byref(s), _C.byref(n)))
        return int(s.value)

    def levenshtein_k(self, k, costs=LEVENSHTEIN_COSTS):
        c = _costs(costs)._c()
        out = np.empty(self.n, dtype=np.uint32)
        _raise(_n.lib().ta_sharded_pairs_levenshtein_k(self._h, int(k) & 0xFFFFFFFF, _C.byref(c), out.ctypes.data))
        return out

    def levenshtein_exp(self, costs=LEVENSHTEIN_COSTS):
        c = _costs(costs)._c()
        out = np.empty(self.n, dtype=np.uint32)
        _raise(_n.lib().ta_sharded_pairs_levenshtein_exp(self._h, _C.byref(c), out.ctypes.data))
        return out

    def hamming(self):
        out = np.empty(self.n, dtype=np.uint32)
        _raise(_n.lib().ta_sharded_pairs_hamming(self._h, out.ctypes.data))
        return out

    def time_levenshtein_k(self, k, costs=LEVENSHTEIN_COSTS, steps=1):
        """`steps` passes back to back on every device -> the slowest shard's device time in ms (HIP events)."""
        c = _costs(costs)._c()
        ms = _C.c_float(0)
        _raise(_n.lib().ta_sharded_pairs_time_levenshtein_k(self._h, int(k) & 0xFFFFFFFF, _C.byref(c), int(steps), _C.byref(ms)))
        return float(ms.value)

    def close(self):
        if self._h:
            _n.lib().ta_sharded_pairs_free(self._h)
            self._h = _C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _take(p, n):
    try:
        return [Match(int(p[i].start), int(p[i].end), int(p[i].k)) for i in range(n.value)]
    finally:
        _n.lib().ta_free(p)


class ShardedHaystack:
    """A haystack uploaded once and kept resident as contiguous shards with `overlap` bytes of their neighbours on either side."""

    def __init__(self, haystack, overlap=4096, n_shards=0):
        buf = np.frombuffer(haystack, dtype=np.uint8) if not isinstance(haystack, np.ndarray) else np.ascontiguousarray(haystack, dtype=np.uint8)
        self.len = int(buf.size)
        self._h = _C.c_void_p()
        _raise(_n.lib().ta_sharded_haystack_upload(_C.c_void_p(buf.ctypes.data if buf.size else 0), self.len, int(overlap), int(n_shards), _C.byref(self._h)))

    @property
    def n_shards(self):
        s, n = _C.c_size_t(0), _C.c_size_t(0)
        _raise(_n.lib().ta_sharded_haystack_shards(self._h, _C.byref(s), _C.byref(n)))
        return int(s.value)

    def levenshtein_search(self, needle, k, search_type=SearchType.Best, costs=LEVENSHTEIN_COSTS):
        needle = bytes(needle)
        c = _costs(costs)._c()
        p, n = _C.POINTER(_n.MatchC)(), _C.c_size_t(0)
        _raise(_n.lib().ta_sharded_haystack_levenshtein_search(self._h, needle, len(needle), int(k) & 0xFFFFFFFF, int(search_type), _C.byref(c), _C.byref(p), _C.byref(n)))
        return _take(p, n)

    def hamming_search(self, needle, k, search_type=SearchType.Best):
        needle = bytes(needle)
        p, n = _C.POINTER(_n.MatchC)(), _C.c_size_t(0)
        _raise(_n.lib().ta_sharded_haystack_hamming_search(self._h, needle, len(needle), int(k) & 0xFFFFFFFF, int(search_type), _C.byref(p), _C.byref(n)))
        return _take(p, n)

    def close(self):
        if self._h:
            _n.lib().ta_sharded_haystack_free(self._h)
            self._h = _C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
