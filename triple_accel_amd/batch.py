"""Batch entry points on device-resident data (torch tensors carry the HBM buffers; the
arithmetic is the HIP kernels behind the C ABI).  New surface with no reference analogue:
N = 1 equals the single-call functions."""
import ctypes as _C

import torch

from . import _native as _n
from . import EditCosts, LEVENSHTEIN_COSTS, _costs, _raise

SLACK = 16


class Strings:
    """A batch side: CSR (blob + n+1 int64 offsets) or strided (fixed length) uint8 data in HBM."""

    def __init__(self, blob, off=None, stride=0, length=0, max_len=0, n=None):
        assert blob.dtype == torch.uint8 and blob.is_cuda and blob.is_contiguous()
        self.blob, self.off, self.stride, self.length, self.max_len = blob, off, stride, length, max_len
        if off is not None:
            assert off.dtype == torch.int64 and off.is_cuda and off.is_contiguous()
            self.n = off.numel() - 1
        else:
            self.n = n if n is not None else (blob.numel() - SLACK) // max(stride, 1)

    def __setattr__(self, name, value):
        # the C view is built once per batch side (_c); changing a field drops it
        if name in ("blob", "off", "stride", "length", "max_len"):
            self.__dict__.pop("_cview", None)
            self.__dict__.pop("_cref", None)
        object.__setattr__(self, name, value)

    @classmethod
    def from_list(cls, strings, device="cuda"):
        import numpy as np
        lens = np.fromiter((len(s) for s in strings), dtype=np.int64, count=len(strings))
        off = np.zeros(len(strings) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        blob = np.zeros(int(off[-1]) + SLACK, dtype=np.uint8)
        if off[-1]:
            blob[:int(off[-1])] = np.frombuffer(b"".join(bytes(s) for s in strings), dtype=np.uint8)
        return cls(torch.from_numpy(blob).to(device), torch.from_numpy(off).to(device),
                   max_len=int(lens.max()) if len(strings) else 0)

    @classmethod
    def from_fixed(cls, array_2d, device="cuda"):
        """(n, len) uint8 array/tensor -> strided batch (stride == len) with read slack."""
        t = torch.as_tensor(array_2d, dtype=torch.uint8)
        n, length = t.shape
        blob = torch.zeros(n * length + SLACK, dtype=torch.uint8, device=device)
        blob[: n * length] = t.reshape(-1).to(device)
        return cls(blob, None, stride=length, length=length, n=n)

    def longest(self):
        """an upper bound on the length of any string of the side: `length` (strided), `max_len` when the caller gave it, else
        measured from the offsets (a CSR side built as Strings(blob, off) has max_len = 0 = "let the library measure it")"""
        if self.off is None:
            return int(self.length)
        if not self.max_len and self.n > 0:
            self.max_len = int((self.off[1:] - self.off[:-1]).max().item())
        return int(self.max_len)

    def _c(self):
        """the C view (built once: the tensors of a batch side do not change)"""
        c = self.__dict__.get("_cview")
        if c is None:
            c = self._cview = _n.StringsC(self.blob.data_ptr(), 0 if self.off is None else self.off.data_ptr(),
                                          self.stride, self.length, self.max_len)
            self._cref = _C.byref(c)
        return c

    def _ref(self):
        self._c()
        return self._cref


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """the current stream's handle (the raw getter skips building a torch.cuda.Stream object per call)"""
    if _raw_stream is not None:
        return _C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return _C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _out(n, device):
    return torch.empty(n, dtype=torch.int32, device=device)


def levenshtein_k_batch(a: Strings, b: Strings, k, costs=LEVENSHTEIN_COSTS, out=None, alphabet=None):
    """out[i] = levenshtein_simd_k_with_opts(a_i, b_i, k, false, costs) as int32 (-1 == None).
    alphabet: the byte values the strings are written in -- up to four (b"ACGT") or up to 32 (IUPAC codes, the amino acids'
    letters): the small-alphabet kernels (same answers; pairs with other bytes are still answered, by the general kernel)."""
    assert a.n == b.n
    out = _out(a.n, a.blob.device) if out is None else out
    cc = _costs(costs)._c()
    if alphabet is not None:
        alphabet = bytes(alphabet)
        rc = _n.lib().ta_levenshtein_k_batch_alphabet(a._ref(), b._ref(), a.n, k, _C.byref(cc), alphabet, len(alphabet), out.data_ptr(), _stream())
        if rc:
            _raise(rc)
        return out
    rc = _n.lib().ta_levenshtein_k_batch(a._ref(), b._ref(), a.n, k, _C.byref(cc), out.data_ptr(), _stream())
    if rc:
        _raise(rc)
    return out


def levenshtein_exp_batch(a: Strings, b: Strings, costs=LEVENSHTEIN_COSTS, out=None):
    assert a.n == b.n
    out = _out(a.n, a.blob.device) if out is None else out
    ca, cb, cc = a._c(), b._c(), _costs(costs)._c()
    _raise(_n.lib().ta_levenshtein_exp_batch(_C.byref(ca), _C.byref(cb), a.n, _C.byref(cc), out.data_ptr(), _stream()))
    return out


_EDIT_NAMES = ("Match", "Mismatch", "AGap", "BGap", "Transpose")


def levenshtein_trace_batch(a: Strings, b: Strings, k, costs=LEVENSHTEIN_COSTS, cap=None, out=None, edits=None, n_edits=None):
    """levenshtein_simd_k_with_opts(a_i, b_i, k, trace_on=True, costs) for every pair, on the device: -> (out, edits, n_edits) with
    out[i] = distance (int32, -1 == None), n_edits[i] = runs of pair i's script, edits[i, t] = (edit type, count) of run t (int64 pairs;
    edit types as triple_accel_amd.EditType).  cap = runs kept per pair (default 2 k + 1: every script of cost <= k fits)."""
    assert a.n == b.n
    n, dev = a.n, a.blob.device
    if cap is None:
        cap = min(2 * int(k) + 1, 2 * max(a.longest(), b.longest(), 1) + 1)
    out = _out(n, dev) if out is None else out
    edits = torch.empty((n, cap, 2), dtype=torch.int64, device=dev) if edits is None else edits
    n_edits = torch.empty(n, dtype=torch.int32, device=dev) if n_edits is None else n_edits
    cc = _costs(costs)._c()
    rc = _n.lib().ta_levenshtein_trace_batch(a._ref(), b._ref(), n, k, _C.byref(cc), out.data_ptr(), edits.data_ptr(), n_edits.data_ptr(),
                                              cap, _stream())
    if rc:
        _raise(rc)
    return out, edits, n_edits


def edits_to_lists(edits, n_edits, allow_cut=False):
    """the device result of levenshtein_trace_batch as Python lists [(name, count), ...] per pair (host copy).  A script that did
    not fit the `cap` records of its pair was cut on the device (n_edits says how long it is): that is an error here, not a
    silently shorter script, unless the caller asks for the cut scripts (allow_cut)."""
    e, ne = edits.cpu().numpy(), n_edits.cpu().numpy()
    cut = [i for i in range(len(ne)) if int(ne[i]) > e.shape[1]]
    if cut and not allow_cut:
        raise ValueError("levenshtein_trace_batch: %d script(s) longer than cap = %d runs (first: pair %d with %d runs); pass a larger cap"
                         % (len(cut), e.shape[1], cut[0], int(ne[cut[0]])))
    return [[(_EDIT_NAMES[int(e[i, t, 0]) & 0xFFFFFFFF], int(e[i, t, 1])) for t in range(min(int(ne[i]), e.shape[1]))] for i in range(len(ne))]


def levenshtein_trace_batch_packed(a: Strings, b: Strings, k, costs=LEVENSHTEIN_COSTS, cap=None, out=None, packed=None, n_edits=None):
    """levenshtein_trace_batch with PACKED records (ta_levenshtein_trace_batch_packed): -> (out, packed, n_edits); packed is (n, cap) int32,
    one word per run -- (edit type << 29) | count -- pair i's script the min(n_edits[i], cap) words at the END of row i, front to back (words in
    front of a script are not written).  A quarter of the bytes of the (n, cap, 2) int64 form; packed_to_lists expands it on the host."""
    assert a.n == b.n
    n, dev = a.n, a.blob.device
    if cap is None:
        cap = min(2 * int(k) + 1, 2 * max(a.longest(), b.longest(), 1) + 1)
    out = _out(n, dev) if out is None else out
    packed = torch.empty((n, cap), dtype=torch.int32, device=dev) if packed is None else packed
    n_edits = torch.empty(n, dtype=torch.int32, device=dev) if n_edits is None else n_edits
    cc = _costs(costs)._c()
    rc = _n.lib().ta_levenshtein_trace_batch_packed(a._ref(), b._ref(), n, k, _C.byref(cc), out.data_ptr(), packed.data_ptr(), n_edits.data_ptr(),
                                                     cap, _stream())
    if rc:
        _raise(rc)
    return out, packed, n_edits


def packed_to_lists(packed, n_edits, allow_cut=False):
    """the result of levenshtein_trace_batch_packed as Python lists [(name, count), ...] per pair (host copy); a script longer than the row
    is an error unless allow_cut (then its LAST cap runs are returned)."""
    import numpy as np
    p, ne = packed.cpu().numpy().view(np.uint32), n_edits.cpu().numpy()
    cap = p.shape[1]
    cut = [i for i in range(len(ne)) if int(ne[i]) > cap]
    if cut and not allow_cut:
        raise ValueError("levenshtein_trace_batch_packed: %d script(s) longer than cap = %d runs (first: pair %d with %d runs); pass a larger cap"
                         % (len(cut), cap, cut[0], int(ne[cut[0]])))
    return [[(_EDIT_NAMES[int(w) >> 29], int(w) & 0x1FFFFFFF) for w in p[i, cap - min(int(ne[i]), cap):]] for i in range(len(ne))]


def hamming_batch(a: Strings, b: Strings, out=None):
    assert a.n == b.n
    out = _out(a.n, a.blob.device) if out is None else out
    rc = _n.lib().ta_hamming_batch(a._ref(), b._ref(), a.n, out.data_ptr(), _stream())
    if rc:
        _raise(rc)
    return out


# ---------------------------------------------------------------- search on a haystack shard resident in HBM
def _hits_to_numpy(hits_t, count):
    """(start, end, k) rows sorted by end.  An end position is reported at most once per call, so `end` alone orders
    the hits; the sort runs on the device and one copy brings the rows to the host."""
    import numpy as np
    if count == 0:
        return np.empty((0, 3), dtype=np.int64)
    v = hits_t[: count * 3].view(-1, 3)                                # (start, end, k|pad) as int64 triples
    out = v[torch.argsort(v[:, 1])].cpu().numpy()
    out[:, 2] &= 0xFFFFFFFF
    return out


import threading as _threading

_hit_bufs = _threading.local()


def _hit_buffer(device, cap):
    """Reused device buffer for hit records (24 B each), one per calling thread (concurrent callers' kernels write their hits into it)."""
    key = (str(device), cap)
    bufs = _hit_bufs.__dict__.setdefault("bufs", {})
    if key not in bufs:
        bufs.clear()
        bufs[key] = torch.empty(cap * 3, dtype=torch.int64, device=device)
    return bufs[key]


def levenshtein_search_dev(needle, haystack, k, costs=LEVENSHTEIN_COSTS, anchored=False, base=0, emit_from=0, cap=None):
    """All-mode hits of levenshtein_search_simd_with_opts over a uint8 CUDA tensor (with >= 16 B of read slack
    after `length`): int64 array of rows (start, end, k) sorted by end.  `base` offsets the positions,
    hits with end <= emit_from are suppressed (left-halo positions of a sharded haystack)."""
    hay, length = haystack if isinstance(haystack, tuple) else (haystack, haystack.numel() - SLACK)
    assert hay.dtype == torch.uint8 and hay.is_cuda and hay.is_contiguous()
    needle = bytes(needle)
    cap = cap or min(length + 2, 1 << 22)
    hits = _hit_buffer(hay.device, cap)
    count = _C.c_uint64()
    cc = _costs(costs)._c()
    rc = _n.lib().ta_levenshtein_search_dev(needle, len(needle), hay.data_ptr(), length, k, _C.byref(cc), int(anchored),
                                            base, emit_from, hits.data_ptr(), cap, _C.byref(count), _stream())
    _raise(rc)
    return _hits_to_numpy(hits, int(count.value))


def levenshtein_search_best_dev(needle, haystack, k, costs=LEVENSHTEIN_COSTS, base=0, emit_from=0, cap=None):
    """The hits of levenshtein_search_dev that can survive the Best fold (those with the smallest k), selected on the
    device: int64 rows (start, end, k) sorted by end -- feed them to dist.fold_best.  Only these few records leave HBM."""
    import numpy as np
    hay, length = haystack if isinstance(haystack, tuple) else (haystack, haystack.numel() - SLACK)
    assert hay.dtype == torch.uint8 and hay.is_cuda and hay.is_contiguous()
    needle = bytes(needle)
    cap = cap or min(length + 2, 1 << 22)
    hits = _hit_buffer(hay.device, cap)
    count = _C.c_uint64()
    cc = _costs(costs)._c()
    out = _C.POINTER(_n.MatchC)()
    n_out = _C.c_size_t()
    rc = _n.lib().ta_levenshtein_search_best_dev(needle, len(needle), hay.data_ptr(), length, k, _C.byref(cc), base, emit_from,
                                                 hits.data_ptr(), cap, _C.byref(count), _C.byref(out), _C.byref(n_out), _stream())
    _raise(rc)
    rows = np.empty((n_out.value, 3), dtype=np.int64)
    for i in range(n_out.value):
        rows[i] = (out[i].start, out[i].end, out[i].k)
    _n.lib().ta_free(out)
    return rows


def levenshtein_search_first_dev(needle, haystack, k, costs=LEVENSHTEIN_COSTS, base=0):
    """The first All-mode hit (smallest end) of a resident haystack shard, or None: the shard is searched window by window and the
    scan stops at the first window that holds a hit (`.next()` on the reference's lazy iterator, src/levenshtein.rs:2282-2420)."""
    hay, length = haystack if isinstance(haystack, tuple) else (haystack, haystack.numel() - SLACK)
    assert hay.dtype == torch.uint8 and hay.is_cuda and hay.is_contiguous()
    needle = bytes(needle)
    m, found = _n.MatchC(), _C.c_int()
    cc = _costs(costs)._c()
    _raise(_n.lib().ta_levenshtein_search_first_dev(needle, len(needle), hay.data_ptr(), length, k, _C.byref(cc), base,
                                                    _C.byref(m), _C.byref(found), _stream()))
    return (int(m.start), int(m.end), int(m.k)) if found.value else None


def hamming_search_dev(needle, haystack, k, base=0, cap=None):
    hay, length = haystack if isinstance(haystack, tuple) else (haystack, haystack.numel() - SLACK)
    needle = bytes(needle)
    cap = cap or min(length + 2, 1 << 22)
    hits = _hit_buffer(hay.device, cap)
    # the hits on the host, sorted by end: the few hits of an ordinary search arrive through pinned memory with the call's one
    # stream synchronisation (ta_hamming_search_dev_sorted; the count + a device-side sort + a copy cost 0.11 ms per call)
    import numpy as np
    out = _C.POINTER(_n.MatchC)()
    n_out = _C.c_size_t()
    rc = _n.lib().ta_hamming_search_dev_sorted(needle, len(needle), hay.data_ptr(), length, k, base, hits.data_ptr(), cap,
                                               _C.byref(out), _C.byref(n_out), _stream())
    _raise(rc)
    n = n_out.value
    if n == 0:
        return np.empty((0, 3), dtype=np.int64)
    raw = np.ctypeslib.as_array(_C.cast(out, _C.POINTER(_C.c_uint64)), shape=(n, 3)).astype(np.int64)     # (start, end, k | pad << 32)
    _n.lib().ta_free(out)
    raw[:, 2] &= 0xFFFFFFFF
    return raw


def haystack_tensor(data, device="cuda"):
    """bytes / uint8 array -> (CUDA tensor with read slack, length)."""
    import numpy as np
    arr = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else np.asarray(data, dtype=np.uint8)
    t = torch.zeros(arr.size + SLACK, dtype=torch.uint8, device=device)
    t[: arr.size] = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
    return t, arr.size
