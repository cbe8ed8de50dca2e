"""triple_accel_amd -- MI355X-native edit-distance engine behind triple_accel's public API.

Host-side mirror of the reference's public function set (triple_accel v0.4.0, src/lib.rs:126-127
and the `levenshtein` / `hamming` modules): same names, argument order and meaning, with
`Option::None` -> `None`, `panic!` -> `PanicError`, `Box<dyn Iterator<Item=Match>>` -> a Python
iterator of `Match`.  All arithmetic runs in HIP kernels on gfx950 through the C ABI of
include/triple_accel_amd.h (libtriple_accel_amd.so); there is no CPU fallback.

The Rust shim a maintainer would drop into the reference crate is shown in INTEGRATION.md.
"""
import ctypes as _C
from collections import namedtuple as _nt

from . import _native as _n
from ._native import TripleAccelError  # noqa: F401


class PanicError(AssertionError):
    """Where the reference panics (assert!/panic!), this is raised."""


# src/lib.rs:135-142
Match = _nt("Match", ["start", "end", "k"])
# src/lib.rs:148-165
Edit = _nt("Edit", ["edit", "count"])


class EditType:
    Match, Mismatch, AGap, BGap, Transpose = "Match", "Mismatch", "AGap", "BGap", "Transpose"


class SearchType:  # src/lib.rs:171-174
    All, Best = 0, 1


class EditCosts:
    """src/levenshtein.rs:20-71.  `EditCosts(mismatch, gap, start_gap, transpose_or_None)` == `EditCosts::new`."""
    __slots__ = ("mismatch_cost", "gap_cost", "start_gap_cost", "transpose_cost")

    def __init__(self, mismatch_cost, gap_cost, start_gap_cost, transpose_cost=None):
        c = _n.EditCostsC()
        rc = _n.lib().ta_edit_costs_new(mismatch_cost, gap_cost, start_gap_cost, transpose_cost is not None,
                                        transpose_cost or 0, _C.byref(c))
        if rc == _n.TA_ERR_BAD_COSTS:
            raise PanicError("invalid EditCosts (src/levenshtein.rs:44-52)")
        _n.check(rc)
        self.mismatch_cost, self.gap_cost, self.start_gap_cost = mismatch_cost, gap_cost, start_gap_cost
        self.transpose_cost = transpose_cost

    new = classmethod(lambda cls, *a: cls(*a))

    def _c(self):
        t = self.transpose_cost
        return _n.EditCostsC(self.mismatch_cost, self.gap_cost, self.start_gap_cost, 0 if t is None else 1, t or 0)

    def __repr__(self):
        return "EditCosts(%d, %d, %d, %r)" % (self.mismatch_cost, self.gap_cost, self.start_gap_cost, self.transpose_cost)


LEVENSHTEIN_COSTS = EditCosts(1, 1, 0, None)   # src/levenshtein.rs:76-81
RDAMERAU_COSTS = EditCosts(1, 1, 0, 1)         # src/levenshtein.rs:84-89


_costs_cache = {}


def _costs(c):
    if isinstance(c, EditCosts):
        return c
    c = tuple(c)
    e = _costs_cache.get(c)
    if e is None:                          # validated once per distinct tuple
        if len(_costs_cache) > 256:
            _costs_cache.clear()
        e = _costs_cache[c] = EditCosts(*c)
    return e


def _raise(rc):
    if rc == _n.TA_ERR_LEN_MISMATCH:
        raise PanicError("assertion failed: a.len() == b.len()")
    if rc == _n.TA_ERR_NULL_BYTE:
        raise PanicError("No zero/null bytes allowed in the string!")
    if rc == _n.TA_ERR_BAD_COSTS:
        raise PanicError("invalid EditCosts")
    if rc == _n.TA_ERR_DIV_ZERO:
        raise PanicError("attempt to divide by zero")
    if rc == _n.TA_ERR_UNSUPPORTED:
        raise NotImplementedError("triple_accel_amd: not on the GPU path (" + _n.lib().ta_last_error().decode() + ")")
    _n.check(rc)


def _u32(fn, *args):
    out = _C.c_uint32()
    _raise(fn(*args, _C.byref(out)))
    return None if out.value == _n.NONE else int(out.value)


def _b(x):
    """bytes-like only: `bytes(5)` would silently turn an int into five NUL bytes."""
    if isinstance(x, bytes):
        return x
    if isinstance(x, (int, str)):
        raise TypeError("expected a bytes-like string, got %s" % type(x).__name__)
    try:                                   # anything with the buffer protocol: bytearray, memoryview, ndarray, array.array, mmap, ctypes arrays
        return bytes(memoryview(x))
    except TypeError:
        raise TypeError("expected a bytes-like string, got %s" % type(x).__name__) from None


def _k(k):
    """k is a u32 in the reference; ctypes would truncate anything else silently."""
    k = int(k)
    if not 0 <= k <= 0xFFFFFFFF:
        raise OverflowError("k must fit a u32, got %d" % k)
    return k


# ---------------------------------------------------------------- hamming (src/hamming.rs)
def hamming(a, b):
    """src/hamming.rs:390"""
    a, b = _b(a), _b(b)
    return _u32(_n.lib().ta_hamming, a, len(a), b, len(b))


def _matches(fn, *args):
    mp = _C.POINTER(_n.MatchC)()
    cnt = _C.c_size_t()
    _raise(fn(*args, _C.byref(mp), _C.byref(cnt)))
    try:
        out = [Match(int(mp[i].start), int(mp[i].end), int(mp[i].k)) for i in range(cnt.value)]
    finally:
        if mp:
            _n.lib().ta_free(mp)
    return iter(out)


def hamming_search_simd_with_opts(needle, haystack, k, search_type):
    """src/hamming.rs:454"""
    needle, haystack = _b(needle), _b(haystack)
    return _matches(_n.lib().ta_hamming_search_simd_with_opts, needle, len(needle), haystack, len(haystack), _k(k),
                    search_type)


def hamming_search_simd(needle, haystack):
    """src/hamming.rs:422"""
    needle, haystack = _b(needle), _b(haystack)
    return _matches(_n.lib().ta_hamming_search, needle, len(needle), haystack, len(haystack))


hamming_search = hamming_search_simd   # src/hamming.rs:588


def hamming_search_naive_with_opts(needle, haystack, k, search_type):
    """src/hamming.rs:96 -- the scalar routine's contract: NUL bytes are fine, an empty needle matches everywhere."""
    needle, haystack = _b(needle), _b(haystack)
    return _matches(_n.lib().ta_hamming_search_naive_with_opts, needle, len(needle), haystack, len(haystack), _k(k),
                    search_type)


def hamming_search_naive(needle, haystack):
    """src/hamming.rs:70"""
    needle = _b(needle)
    return hamming_search_naive_with_opts(needle, haystack, (len(needle) >> 1) + (len(needle) & 1), SearchType.Best)



# ---------------------------------------------------------------- levenshtein (src/levenshtein.rs)
_EDIT_NAMES = [EditType.Match, EditType.Mismatch, EditType.AGap, EditType.BGap, EditType.Transpose]


def _trace(fn, *args):
    out = _C.c_uint32()
    ep = _C.POINTER(_n.EditC)()
    cnt = _C.c_size_t()
    _raise(fn(*args, _C.byref(out), _C.byref(ep), _C.byref(cnt)))
    try:
        edits = [Edit(_EDIT_NAMES[ep[i].edit], int(ep[i].count)) for i in range(cnt.value)]
    finally:
        if ep:
            _n.lib().ta_free(ep)
    return (None, None) if out.value == _n.NONE else (int(out.value), edits)


def levenshtein_simd_k_with_opts(a, b, k, trace_on, costs):
    """src/levenshtein.rs:714 -> None | (distance, None | [Edit])."""
    a, b = _b(a), _b(b)
    if trace_on:
        d, edits = _trace(_n.lib().ta_levenshtein_trace, a, len(a), b, len(b), _k(k), _C.byref(_costs(costs)._c()))
        return None if d is None else (d, edits)
    d = _u32(_n.lib().ta_levenshtein_simd_k_with_opts, a, len(a), b, len(b), _k(k), 0, _C.byref(_costs(costs)._c()))
    return None if d is None else (d, None)


def levenshtein_simd_k(a, b, k):
    """src/levenshtein.rs:677"""
    a, b = _b(a), _b(b)
    return _u32(_n.lib().ta_levenshtein_simd_k, a, len(a), b, len(b), _k(k))


def _dist(name):
    def f(a, b):
        a, b = _b(a), _b(b)
        return _u32(getattr(_n.lib(), name), a, len(a), b, len(b))
    return f


levenshtein = _dist("ta_levenshtein")            # src/levenshtein.rs:1397
rdamerau = _dist("ta_rdamerau")                  # :1419
levenshtein_exp = _dist("ta_levenshtein_exp")    # :1445
rdamerau_exp = _dist("ta_rdamerau_exp")          # :1516


def levenshtein_exp_with_opts(a, b, trace_on, costs):
    """src/levenshtein.rs:1480 -> (distance, None | [Edit])"""
    a, b = _b(a), _b(b)
    if trace_on:
        return _trace(_n.lib().ta_levenshtein_exp_trace, a, len(a), b, len(b), _C.byref(_costs(costs)._c()))
    d = _u32(_n.lib().ta_levenshtein_exp_with_opts, a, len(a), b, len(b), 0, _C.byref(_costs(costs)._c()))
    return (d, None)


def levenshtein_search_first(needle, haystack, k, costs=LEVENSHTEIN_COSTS, anchored=False):
    """The first element of the All-mode result -- `.next()` on the reference's lazy iterator (src/levenshtein.rs:2282-2420;
    tests/basic_tests.rs:628-632) -- or None; scans (and uploads) the haystack only as far as that match."""
    needle, haystack = _b(needle), _b(haystack)
    m, found = _n.MatchC(), _C.c_int()
    _raise(_n.lib().ta_levenshtein_search_first(needle, len(needle), haystack, len(haystack), _k(k), _C.byref(_costs(costs)._c()),
                                               int(bool(anchored)), _C.byref(m), _C.byref(found)))
    return Match(int(m.start), int(m.end), int(m.k)) if found.value else None


_LAZY_SEARCH_FROM = 1 << 20     # All-mode searches over haystacks this long hand their first match out before scanning the rest


class _LazyMatches:
    """All-mode result whose first element comes from ta_levenshtein_search_first; the full search runs when a second one is
    asked for (the same sequence as the eager list, element for element)."""

    def __init__(self, needle, haystack, k, costs, anchored):
        # what the eager search reports at the call is reported at the call here too (not inside next()): needle length, device
        if len(needle) > 65535:
            _raise(_n.TA_ERR_ARG)
        if _n.lib().ta_device_count() <= 0:
            _raise(_n.TA_ERR_HIP)
        self._args, self._state, self._rest = (needle, haystack, k, costs, anchored), 0, None

    def __iter__(self):
        return self

    def __next__(self):
        needle, haystack, k, costs, anchored = self._args
        if self._state == 0:
            self._state = 1
            first = levenshtein_search_first(needle, haystack, k, costs, anchored)
            if first is None:
                self._state = 2
                self._rest = iter(())
                raise StopIteration
            return first
        if self._state == 1:
            self._state = 2
            # (ta_levenshtein_search_resume: the bytes the first call uploaded stay on the device, only the rest of the haystack travels;
            # `haystack` is an immutable bytes object held by this iterator: the same pointer, the same contents)
            self._rest = _matches(_n.lib().ta_levenshtein_search_resume, needle, len(needle), haystack, len(haystack), k,
                                  _C.byref(costs._c()), int(bool(anchored)))
            next(self._rest)                                  # the element already handed out
        return next(self._rest)


def levenshtein_search_simd_with_opts(needle, haystack, k, search_type, costs, anchored):
    """src/levenshtein.rs:1911"""
    needle, haystack = _b(needle), _b(haystack)
    if search_type == SearchType.All and len(haystack) >= _LAZY_SEARCH_FROM and len(needle):
        cc = _costs(costs)
        _raise(_n.lib().ta_edit_costs_check_search(_C.byref(cc._c())))     # the panics of the call itself stay eager (:1965)
        return _LazyMatches(needle, haystack, _k(k), cc, anchored)
    return _matches(_n.lib().ta_levenshtein_search_simd_with_opts, needle, len(needle), haystack, len(haystack), _k(k),
                    search_type, _C.byref(_costs(costs)._c()), int(bool(anchored)))


def levenshtein_search_simd(needle, haystack):
    """src/levenshtein.rs:1866"""
    needle, haystack = _b(needle), _b(haystack)
    return _matches(_n.lib().ta_levenshtein_search, needle, len(needle), haystack, len(haystack))


levenshtein_search = levenshtein_search_simd   # src/levenshtein.rs:2508


def levenshtein_select(a_len, b_len, k, costs=LEVENSHTEIN_COSTS):
    """The dispatcher arithmetic (src/levenshtein.rs:731-791): (max_k, unit_k, cell_bits, ref_lanes)."""
    s = _n.LevSelectC()
    _raise(_n.lib().ta_levenshtein_select(a_len, b_len, _k(k), _C.byref(_costs(costs)._c()), _C.byref(s)))
    return (s.max_k, s.unit_k, s.cell_bits, s.ref_lanes)


def last_launch_info():
    li = _n.LaunchInfoC()
    _n.check(_n.lib().ta_last_launch_info(_C.byref(li)))
    return {n: int(getattr(li, n)) for n, _ in _n.LaunchInfoC._fields_}


def levenshtein_simd_k_with_opts_many(pairs, k, costs=None, flush_every=1 << 16):
    """[levenshtein_simd_k_with_opts(a, b, k, False, costs)[0] ... for (a, b) in pairs] -- what a caller's LOOP over the drop-in function
    computes, answered by one batch pass per `flush_every` pairs (Queue): a single call costs a kernel launch (22-25 us for a 256-byte
    pair against ~2 us on a host core), a queued pair 1.0-1.8 us.  Returns distances (None where the distance exceeds k)."""
    q = Queue(k, costs)
    out = []
    try:
        pending = 0
        for a, b in pairs:
            q.push(a, b)
            pending += 1
            if pending >= flush_every:
                out.extend(q.flush())
                pending = 0
        if pending:
            out.extend(q.flush())
    finally:
        q.close()
    return out


def levenshtein_many(pairs):
    """[levenshtein(a, b) for (a, b) in pairs] through the queue (see levenshtein_simd_k_with_opts_many)."""
    return levenshtein_simd_k_with_opts_many(pairs, 0xFFFFFFFF, LEVENSHTEIN_COSTS)


def last_kernel_name():
    """The dominant kernel of this thread's last pass, as a profiler prints it (without `void ta::` and the parameter list)."""
    return _n.lib().ta_last_kernel_name().decode()


class Queue:
    """Pairs produced one at a time, answered together (include/triple_accel_amd.h, ta_queue_*): push() copies a pair and returns its
    ticket, flush() runs ONE batch pass and returns the answers in push order (None where the distance exceeds k)."""

    def __init__(self, k, costs=None):
        self._q = _C.c_void_p()
        cc = _costs(costs if costs is not None else LEVENSHTEIN_COSTS)._c()
        _raise(_n.lib().ta_queue_create(_k(k), _C.byref(cc), _C.byref(self._q)))

    def push(self, a, b):
        a, b = _b(a), _b(b)
        t = _C.c_size_t()
        _raise(_n.lib().ta_queue_push(self._q, a, len(a), b, len(b), _C.byref(t)))
        return int(t.value)

    def flush(self):
        res, n = _C.POINTER(_C.c_uint32)(), _C.c_size_t()
        _raise(_n.lib().ta_queue_flush(self._q, _C.byref(res), _C.byref(n)))
        return [None if res[i] == _n.NONE else int(res[i]) for i in range(n.value)]

    def close(self):
        if self._q:
            _n.lib().ta_queue_destroy(self._q)
            self._q = _C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


OPT_EARLY_OUT = 1
OPT_UNIT_PREFILTER = 2


def set_option(option, value):
    """Options of the calling thread (include/triple_accel_amd.h): OPT_EARLY_OUT -- the band kernels of fixed-length unit-cost
    batches stop a wavefront once none of its pairs can end at or below k; OPT_UNIT_PREFILTER -- batches under weighted EditCosts run the
    unit-cost pass first and price only the pairs it could not rule out (both: same answers, data-dependent work)."""
    _n.check(_n.lib().ta_set_option(int(option), int(bool(value))))


def thread_release():
    """Free what the calling thread holds inside the library (its stream, pinned buffers, device scratch).  A worker thread calls
    this before it ends; otherwise those live until process exit."""
    _n.lib().ta_thread_release()


def device_count():
    return int(_n.lib().ta_device_count())


def version():
    return _n.lib().ta_version().decode()


# ---------------------------------------------------------------- host-only front ends and same-result variants
def levenshtein_simd_k_str(a: str, b: str, k):
    """src/levenshtein.rs:641-651: UTF-8 front end.  ASCII goes through as bytes; otherwise every distinct char is
    remapped to one byte (`translate_str`, :609-624) -- None if the two strings use more than 256 distinct chars."""
    if a.isascii() and b.isascii():
        return levenshtein_simd_k(a.encode("ascii"), b.encode("ascii"), k)
    table = {}

    def translate(s):
        out = bytearray()
        for ch in s:
            if ch not in table:
                if len(table) >= 256:
                    return None
                table[ch] = len(table)
            out.append(table[ch])
        return bytes(out)

    ta = translate(a)
    if ta is None:
        return None
    tb = translate(b)
    if tb is None:
        return None
    return levenshtein_simd_k(ta, tb, k)


def alloc_str(length):
    """src/lib.rs:197-205 (16-byte aligned/padded buffer in the reference; alignment is irrelevant here)."""
    return bytearray(length)


def fill_str(dest, src):
    """src/lib.rs:229-235"""
    if len(dest) < len(src):
        raise PanicError("assertion failed: dest.len() >= src.len()")
    dest[: len(src)] = src


# The reference's other Hamming routines (src/hamming.rs:176-367) are CPU tricks with the same result contract as
# `hamming`; on the GPU they are the same kernel.
hamming_simd_parallel = hamming
hamming_simd_movemask = hamming
hamming_words_64 = hamming
hamming_words_128 = hamming


def rdamerau_simd_k(a, b, k):
    """convenience: levenshtein_simd_k_with_opts(a, b, k, false, RDAMERAU_COSTS) -> Option<u32>"""
    r = levenshtein_simd_k_with_opts(a, b, k, False, RDAMERAU_COSTS)
    return None if r is None else r[0]


# ---- the reference's scalar entry points (same result contract as the kernels implement: the scalar path IS the
# bit-exactness target), under their own names so that callers of any reference `pub fn` are unchanged
hamming_naive = hamming                                   # src/hamming.rs:36


def _symbols(a, b):
    """Two sequences of arbitrary equality-comparable items -> byte strings over a shared code table (the generic
    `T: PartialEq` entry points, src/levenshtein.rs:105, :148, :376).  The recurrence only ever compares an item of `a` with an
    item of `b`, so every item that occurs in ONE of the two sequences only can share a single code per side (it never equals
    anything on the other side): the byte kernels serve any pair with at most 254 distinct items COMMON to both.  Byte strings
    go through unchanged when both are; a byte string next to a general sequence is a sequence of ints."""
    if all(isinstance(s, (bytes, bytearray, memoryview)) for s in (a, b)):
        return [bytes(a), bytes(b)]
    a, b = list(a), list(b)
    try:                                   # hashable items: O(n)
        common = set(a) & set(b)
        code = {x: i for i, x in enumerate(common)}
        enc = lambda x: code.get(x)
    except TypeError:                      # only `==` is promised
        common = [x for n, x in enumerate(a) if x in b and x not in a[:n]]
        enc = lambda x: common.index(x) if x in common else None
    if len(common) > 254:
        raise NotImplementedError("triple_accel_amd: more than 254 distinct symbols common to both sequences cannot be mapped "
                                  "onto the byte kernels")
    ea = bytes(254 if (c := enc(x)) is None else c for x in a)      # 254: only in a, 255: only in b
    eb = bytes(255 if (c := enc(x)) is None else c for x in b)
    return [ea, eb]


def levenshtein_naive_with_opts(a, b, trace_on, costs):
    """src/levenshtein.rs:148 -> (distance, None | [Edit])"""
    a, b = _symbols(a, b)
    return levenshtein_simd_k_with_opts(a, b, 0xFFFFFFFF, trace_on, costs)


def levenshtein_naive(a, b):
    """src/levenshtein.rs:105"""
    return levenshtein_naive_with_opts(a, b, False, LEVENSHTEIN_COSTS)[0]


def levenstein_naive_str(a: str, b: str):
    """src/levenshtein.rs:123 (the reference's spelling) -- over chars, not bytes"""
    return levenshtein_naive(list(a), list(b))


def levenshtein_naive_k_with_opts(a, b, k, trace_on, costs):
    """src/levenshtein.rs:376"""
    a, b = _symbols(a, b)
    return levenshtein_simd_k_with_opts(a, b, k, trace_on, costs)


def levenshtein_naive_k(a, b, k):
    """src/levenshtein.rs:342"""
    return levenshtein_simd_k(a, b, k)


def levenshtein_search_naive_with_opts(needle, haystack, k, search_type, costs, anchored):
    """src/levenshtein.rs:1589"""
    return levenshtein_search_simd_with_opts(needle, haystack, k, search_type, costs, anchored)


def levenshtein_search_naive(needle, haystack):
    """src/levenshtein.rs:1549"""
    return levenshtein_search_simd(needle, haystack)
