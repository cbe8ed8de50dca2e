"""ctypes binding of the host emulation of the kernel bodies (tests/emu/libta_emu.so). TESTS ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SAN = os.environ.get("TA_SANITIZED") == "1"        # scripts/run_sanitized.sh: the ASan + UBSan build
_SO = os.path.join(_HERE, "emu", "libta_emu_san.so" if _SAN else "libta_emu.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "emu"), "-j", str(os.cpu_count() or 4), "-s"] + (["SAN=1"] if _SAN else []))
        L = C.CDLL(_SO)
        L.emu_lev_band.restype = C.c_int
        L.emu_lev_band.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                   C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint64,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.emu_lev_bits.restype = C.c_int
        L.emu_lev_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                   C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.emu_lev_widebits.restype = C.c_int
        L.emu_lev_widebits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                       C.c_uint64, C.c_int, C.c_uint32, C.c_void_p]
        L.emu_lev_filter.restype = C.c_int
        L.emu_lev_filter.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint64,
                                     C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        L.emu_ham_search.restype = C.c_int
        L.emu_ham_search.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int,
                                     C.c_void_p, C.c_uint64, C.c_void_p]
        L.emu_lev_widebits_huge.restype = C.c_uint32
        L.emu_lev_widebits_huge.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_int]
        L.emu_lev_widebits_trace.restype = C.c_int
        L.emu_lev_widebits_trace.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.emu_lev_bits_any.restype = C.c_int
        L.emu_lev_bits_any.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                       C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.emu_lev_bits2.restype = C.c_int
        L.emu_lev_bits2.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                    C.c_void_p, C.c_void_p]
        L.emu_lev_one.restype = C.c_int
        L.emu_lev_one.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p]
        L.emu_lev_search.restype = C.c_int
        L.emu_lev_search.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p,
                                     C.c_uint64, C.POINTER(C.c_uint64)]
        L.emu_search_filter_k.restype = C.c_uint32
        L.emu_search_filter_k.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
        L.emu_search_anchored_packed_ok.restype = C.c_int
        L.emu_search_anchored_packed_ok.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.emu_sliced_plan.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
        L.emu_sliced_plan.restype = None
        _lib = L
    return _lib


def pack(strings):
    """list of bytes -> (blob uint8 array with 16 B slack, uint64 offsets)."""
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in strings])
    blob = np.zeros(int(off[-1]) + 16, dtype=np.uint8)
    if off[-1]:
        blob[:int(off[-1])] = np.frombuffer(b"".join(strings), dtype=np.uint8)
    return blob, off


def lev_band_score_applies(costs, force_trans_select=False):
    """what the launcher decides: the score form of the band kernel for these costs?"""
    mc, gc, sg, tc = costs
    trans = 0 if tc is None else (1 if 2 * mc <= 255 + tc and not force_trans_select else 2)
    return bool(lib().emu_lev_score_applies(mc, gc, trans, 0 if tc is None else tc))


def band_line(on):
    """Fixed-length batches in the one-lane-per-pair layout through the LINE form of the DP band kernel's fetch (whole 128-byte lines parked
    in registers; score form), as the launcher does; LDS starts out as 0xA5 garbage."""
    lib().emu_lev_set_line(1 if on else 0)


def lev_band(a_list, b_list, k, costs=(1, 1, 0, None), force_D=0, force_L=0, force_affine=False, force_trans_select=False,
             chunk=0, score=True):
    """-> (list of dist|None, plan dict); chunk = bytes per streamed LDS chunk (0: the planner's choice);
    score=False: the cost form even where the launcher would take the score form"""
    lib().emu_lev_set_chunk(int(chunk))
    lib().emu_lev_set_score(-1 if score else 0)
    n = len(a_list)
    ab, ao = pack(a_list)
    bb, bo = pack(b_list)
    out = np.full(n, 0xDEADBEEF, dtype=np.uint32)
    plan = np.zeros(5, dtype=np.uint32)
    max_len = max([len(x) for x in a_list] + [len(x) for x in b_list] + [0])
    mc, gc, sg, tc = costs
    rc = lib().emu_lev_band(ab.ctypes.data, ao.ctypes.data, bb.ctypes.data, bo.ctypes.data, n, k, mc, gc, sg,
                            0 if tc is None else 1, 0 if tc is None else tc, max_len, force_D, force_L,
                            int(force_affine) | (2 if force_trans_select else 0), out.ctypes.data, plan.ctypes.data)
    if rc:
        raise RuntimeError("emu_lev_band rc=%d" % rc)
    res = [None if int(x) == 0xFFFFFFFF else int(x) for x in out]
    return res, dict(D=int(plan[0]), L=int(plan[1]), PW=int(plan[2]), u=int(plan[3]), o=int(plan[4]))


def lev_bits(a_list, b_list, k, trans=False, force_NA=0, chunk=0, static=0):
    """Bit-parallel band kernel body (unit costs).  static: 0 planner's choice, 1 sliding window, 2 static window, 3 stride-8 window.
    -> (list of dist|None, plan dict)"""
    lib().emu_lev_set_chunk(int(chunk))
    n = len(a_list)
    ab, ao = pack(a_list)
    bb, bo = pack(b_list)
    out = np.full(n, 0xDEADBEEF, dtype=np.uint32)
    plan = np.zeros(4, dtype=np.uint32)
    max_len = max([len(x) for x in a_list] + [len(x) for x in b_list] + [0])
    rc = lib().emu_lev_bits(ab.ctypes.data, ao.ctypes.data, bb.ctypes.data, bo.ctypes.data, n, k, int(bool(trans)),
                            max_len, force_NA, static, out.ctypes.data, plan.ctypes.data)
    lib().emu_lev_set_chunk(0)
    if rc:
        raise RuntimeError("emu_lev_bits rc=%d" % rc)
    res = [None if int(x) == 0xFFFFFFFF else int(x) for x in out]
    return res, dict(NA=int(plan[0]), u=int(plan[1]), Tw=int(plan[2]), static=int(plan[3]) == 1, s8=int(plan[3]) == 3)


def lev_bits_trace(a_list, b_list, u, dists, trans=False, tile=16, fixed=False, packed=0):
    """Batch tracebacks by checkpoints + recomputation (lev_bits_trace_body.h): the run-length script the reference returns --
    [(edit name, count)] per pair, None for a pair whose distance is None (the body leaves the runs last run first: turned round here, as
    the kernel's last step does).  dists: the pass's answers (None = no script); u: the pass's unit_k.  fixed: the strided (fixed-length)
    view of the batch instead of CSR.  packed = c > 0: the packed form (LevBitsTraceParams::packed_cap = c) -- the walk writes the runs
    right-aligned into c words per pair, front to back; a script of more than c runs keeps its LAST c runs."""
    n = len(a_list)
    ab, ao = pack(a_list)
    bb, bo = pack(b_list)
    max_len = max([len(x) for x in a_list] + [len(x) for x in b_list] + [0])
    cap = min(2 * max_len + 1, 2 * u + 2)
    lib().emu_lev_bits_trace_set_packed(int(packed))
    if packed:
        cap = int(packed)
    runs = np.full(n * cap, 0xDEADBEEF, dtype=np.uint32)
    n_runs = np.full(n, 0xDEADBEEF, dtype=np.uint32)
    dist = np.array([0xFFFFFFFF if d is None else d for d in dists], dtype=np.uint32)
    f = lib().emu_lev_bits_trace
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p,
                  C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    if fixed:
        la, lb = len(a_list[0]), len(b_list[0])
        assert all(len(x) == la for x in a_list) and all(len(x) == lb for x in b_list)
        rc = f(ab.ctypes.data, None, la, bb.ctypes.data, None, lb, n, u, int(bool(trans)), tile, dist.ctypes.data, max_len, runs.ctypes.data, cap, n_runs.ctypes.data)
    else:
        rc = f(ab.ctypes.data, ao.ctypes.data, 0, bb.ctypes.data, bo.ctypes.data, 0, n, u, int(bool(trans)), tile, dist.ctypes.data, max_len, runs.ctypes.data, cap, n_runs.ctypes.data)
    lib().emu_lev_bits_trace_set_packed(0)
    if rc:
        raise RuntimeError("emu_lev_bits_trace rc=%d" % rc)
    names = ["Match", "Mismatch", "AGap", "BGap", "Transpose"]
    if packed:
        out = []
        for p in range(n):
            if dists[p] is None:
                assert n_runs[p] == 0 and np.all(runs[p * cap:(p + 1) * cap] == 0xDEADBEEF)
                out.append(None)
                continue
            have = min(int(n_runs[p]), cap)
            assert np.all(runs[p * cap:(p + 1) * cap - have] == 0xDEADBEEF)          # nothing outside the script's words
            ws = [int(w) for w in runs[(p + 1) * cap - have:(p + 1) * cap]]
            out.append((int(n_runs[p]), [(names[w >> 29], w & 0x1FFFFFFF) for w in ws]))
        return out
    out = []
    for p in range(n):
        if dists[p] is None:
            assert n_runs[p] == 0
            out.append(None)
            continue
        nr = int(n_runs[p])
        assert nr <= cap, (p, nr, cap)
        ws = [int(w) for w in runs[p * cap:p * cap + nr]][::-1]
        out.append([(names[w >> 29], w & 0x1FFFFFFF) for w in ws])
    return out


def bits_fixed_chunk(on):
    """Fixed-length batches through the CHUNK form of the fetch (what the launcher picks up to one 128-byte line per string)."""
    lib().emu_bits_set_fixed_chunk(1 if on else 0)


def bits_vline(on):
    """The VLINE form of the fetch (whole 128-byte lines per lane, per-lane geometry and alignment: what the launcher gives CSR batches)
    for every following lev_bits / lev_bits_fixed call; LDS starts out as 0xA5 garbage, bytes outside the blobs read as 0xA5."""
    lib().emu_bits_set_vline(1 if on else 0)


def lev_bits_fixed(a2d, b2d, k, trans=False, force_NA=0, static=0, subset=None):
    """The same body on a fixed-length (strided) batch -- (n, La) and (n, Lb) uint8 arrays -- which takes the COALESCED fetch
    form; subset: optional pair indices (results land at out[pair], other entries stay 0xDEADBEEF -> 'untouched')."""
    a2d, b2d = np.ascontiguousarray(a2d, dtype=np.uint8), np.ascontiguousarray(b2d, dtype=np.uint8)
    n_all, la = a2d.shape
    lb = b2d.shape[1]
    ab = np.concatenate([a2d.reshape(-1), np.zeros(16, dtype=np.uint8)])
    bb = np.concatenate([b2d.reshape(-1), np.zeros(16, dtype=np.uint8)])
    out = np.full(n_all, 0xDEADBEEF, dtype=np.uint32)
    plan = np.zeros(4, dtype=np.uint32)
    sub = None if subset is None else np.ascontiguousarray(subset, dtype=np.uint32)
    n = n_all if sub is None else len(sub)
    rc = lib().emu_lev_bits_any(ab.ctypes.data, None, la, bb.ctypes.data, None, lb, None if sub is None else sub.ctypes.data,
                                n, k, int(bool(trans)), max(la, lb), force_NA, static, out.ctypes.data, plan.ctypes.data)
    if rc:
        raise RuntimeError("emu_lev_bits_any rc=%d" % rc)
    res = ["untouched" if int(x) == 0xDEADBEEF else None if int(x) == 0xFFFFFFFF else int(x) for x in out]
    return res, dict(NA=int(plan[0]), u=int(plan[1]), Tw=int(plan[2]), static=int(plan[3]) == 1, s8=int(plan[3]) == 3)


def bits_set_tune(bits):
    """LevParams::tune for the bit-parallel bodies (2: early out)."""
    lib().emu_bits_set_tune.argtypes = [C.c_uint32]
    lib().emu_bits_set_tune(bits)


def lev_bits2(a2d, b2d, k, trans=False, subset=None):
    """Two pairs per lane (lev_bits2_body.h) on a fixed-length batch; None when the planner declines (band wider than 15)."""
    a2d, b2d = np.ascontiguousarray(a2d, dtype=np.uint8), np.ascontiguousarray(b2d, dtype=np.uint8)
    n_all, la = a2d.shape
    lb = b2d.shape[1]
    ab = np.concatenate([a2d.reshape(-1), np.zeros(16, dtype=np.uint8)])
    bb = np.concatenate([b2d.reshape(-1), np.zeros(16, dtype=np.uint8)])
    out = np.full(n_all, 0xDEADBEEF, dtype=np.uint32)
    plan = np.zeros(3, dtype=np.uint32)
    sub = None if subset is None else np.ascontiguousarray(subset, dtype=np.uint32)
    n = n_all if sub is None else len(sub)
    rc = lib().emu_lev_bits2(ab.ctypes.data, la, bb.ctypes.data, lb, None if sub is None else sub.ctypes.data, n, k, int(bool(trans)),
                             out.ctypes.data, plan.ctypes.data)
    if rc == 1:
        return None, None
    if rc:
        raise RuntimeError("emu_lev_bits2 rc=%d" % rc)
    res = ["untouched" if int(x) == 0xDEADBEEF else None if int(x) == 0xFFFFFFFF else int(x) for x in out]
    return res, dict(NA=int(plan[0]), u=int(plan[1]), Tw=int(plan[2]))


def lev_bitsqw(a2d, b2d, k, alphabet, trans=False, subset=None):
    """The form for alphabets of up to 32 symbols (lev_bitsqw_body.h); same contract as lev_bitsq."""
    return lev_bitsq(a2d, b2d, k, alphabet, trans, subset, entry="emu_lev_bitsqw")


def lev_bitsq(a2d, b2d, k, alphabet, trans=False, subset=None, entry="emu_lev_bitsq"):
    """Small-alphabet form (lev_bitsq_body.h) on a fixed-length batch -> (results with 'untouched' where a pair holds a byte outside
    the alphabet, sorted list of those pairs); (None, None) when the planner declines or the alphabet has no code hash."""
    a2d, b2d = np.ascontiguousarray(a2d, dtype=np.uint8), np.ascontiguousarray(b2d, dtype=np.uint8)
    n_all, la = a2d.shape
    lb = b2d.shape[1]
    ab = np.concatenate([a2d.reshape(-1), np.zeros(16, dtype=np.uint8)])
    bb = np.concatenate([b2d.reshape(-1), np.zeros(16, dtype=np.uint8)])
    out = np.full(n_all, 0xDEADBEEF, dtype=np.uint32)
    bad = np.zeros(n_all + 1, dtype=np.uint32)
    sub = None if subset is None else np.ascontiguousarray(subset, dtype=np.uint32)
    n = n_all if sub is None else len(sub)
    f = getattr(lib(), entry)
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, C.c_uint32,
                  C.c_void_p, C.c_void_p]
    rc = f(ab.ctypes.data, la, bb.ctypes.data, lb, None if sub is None else sub.ctypes.data, n, k, int(bool(trans)), bytes(alphabet),
           len(alphabet), out.ctypes.data, bad.ctypes.data)
    if rc in (1, 3):
        return None, None
    if rc:
        raise RuntimeError("%s rc=%d" % (entry, rc))
    res = ["untouched" if int(x) == 0xDEADBEEF else None if int(x) == 0xFFFFFFFF else int(x) for x in out]
    return res, sorted(int(x) for x in bad[1:1 + int(bad[0])])


def lev_one(a, b, k, trans=False):
    """One pair through the single-pair kernel body (lev_one_body.h).  -> dist | None, or "declined" (band wider than 64)."""
    ab = np.frombuffer(bytes(a) + bytes(32), dtype=np.uint8).copy()
    bb = np.frombuffer(bytes(b) + bytes(32), dtype=np.uint8).copy()
    out = np.full(1, 0xDEADBEEF, dtype=np.uint32)
    rc = lib().emu_lev_one(ab.ctypes.data, len(a), bb.ctypes.data, len(b), k, int(bool(trans)), out.ctypes.data)
    if rc == 1:
        return "declined"
    if rc:
        raise RuntimeError("emu_lev_one rc=%d" % rc)
    return None if int(out[0]) == 0xFFFFFFFF else int(out[0])


def lev_widebits(a_list, b_list, k, trans=False, nwl=2, nwaves=3):
    """Row-blocked bit-parallel kernel body (unit costs, shorter string <= 64 * 32 * nwl bytes).  -> list of dist|None"""
    n = len(a_list)
    ab, ao = pack(a_list)
    bb, bo = pack(b_list)
    out = np.full(n, 0xDEADBEEF, dtype=np.uint32)
    max_len = max([len(x) for x in a_list] + [len(x) for x in b_list] + [0])
    rc = lib().emu_lev_widebits(ab.ctypes.data, ao.ctypes.data, bb.ctypes.data, bo.ctypes.data, n, k, int(bool(trans)),
                                max_len, nwl, nwaves, out.ctypes.data)
    if rc:
        raise RuntimeError("emu_lev_widebits rc=%d" % rc)
    return [None if int(x) == 0xFFFFFFFF else int(x) for x in out]


def lev_search_tiled(needle, haystack, k, costs=(1, 1, 0, None), anchored=False, tile=64, halo=None, packed=False):
    """All-mode hits [(start, end, k)] of the tile function run over a tiled haystack (halo = n + unit_k + 2)."""
    mc, gc, sg, tc = costs
    n = len(needle)
    if halo is None:
        halo = n + max(0, k - sg) // gc + 2
    hay = np.zeros(len(haystack) + 16, dtype=np.uint8)
    hay[:len(haystack)] = np.frombuffer(haystack, dtype=np.uint8)
    h = len(haystack)
    if anchored:
        h = min(h, n + max(0, k - sg) // gc)
        tile, halo = 1 << 40, 0
    lib().emu_search_set_packed(int(packed))
    cap = len(haystack) + 2
    out = np.zeros((cap, 3), dtype=np.uint64)
    cnt = C.c_uint64()
    rc = lib().emu_lev_search(needle, n, hay.ctypes.data, h, k, mc, gc, sg, 0 if tc is None else 1,
                              0 if tc is None else tc, int(anchored), tile, halo, out.ctypes.data, cap, C.byref(cnt))
    if rc:
        raise RuntimeError("emu_lev_search rc=%d" % rc)
    res = []
    for i in range(cnt.value):
        res.append((int(out[i, 0]), int(out[i, 1]), int(out[i, 2] & np.uint64(0xFFFFFFFF))))
    return res


def search_filter_k(k, costs):
    """k' of the superset filter under general EditCosts (srch_filter_k, lev_search_body.h)"""
    mc, gc, sg, tc = costs
    return int(lib().emu_search_filter_k(k, mc, gc, sg, 0 if tc is None else 1, 0 if tc is None else tc))


def lev_filter_blocks(needle, haystack, k, trans=False, tile=256, halo=None, words=0):
    """64-column blocks the bit-parallel filter flags (sorted list of block indices)."""
    n = len(needle)
    if halo is None:
        halo = n + k + 2
    hay = np.zeros(len(haystack) + 16, dtype=np.uint8)
    hay[:len(haystack)] = np.frombuffer(haystack, dtype=np.uint8)
    cap = len(haystack) // 64 + 2
    out = np.zeros(cap, dtype=np.uint64)
    cnt = C.c_uint64()
    rc = lib().emu_lev_filter(needle, n, hay.ctypes.data, len(haystack), k, int(bool(trans)), tile, halo, words,
                              out.ctypes.data, cap, C.byref(cnt))
    if rc:
        raise RuntimeError("emu_lev_filter rc=%d" % rc)
    return sorted(int(x) for x in out[:cnt.value])


def ham_search(needle, haystack, k, tile=512, words=0):
    """All-mode hits [(start, end, k)] of the shift-add hamming_search scan over a tiled haystack."""
    n = len(needle)
    hay = np.zeros(len(haystack) + 16, dtype=np.uint8)
    hay[:len(haystack)] = np.frombuffer(haystack, dtype=np.uint8)
    cap = len(haystack) + 2
    out = np.zeros((cap, 3), dtype=np.uint64)
    cnt = C.c_uint64()
    rc = lib().emu_ham_search(needle, n, hay.ctypes.data, len(haystack), k, tile, words, out.ctypes.data, cap, C.byref(cnt))
    if rc:
        raise RuntimeError("emu_ham_search rc=%d" % rc)
    return [(int(out[i, 0]), int(out[i, 1]), int(out[i, 2] & np.uint64(0xFFFFFFFF))) for i in range(cnt.value)]


def ham_search_bits(needle, haystack, k, tile=256):
    """All-mode hits of the bit-sliced hamming_search form (ham_bits_body.h) over a tiled haystack; None where the form does not apply
    (k >= n, or k needs more than five counter bits)."""
    n = len(needle)
    hay = np.zeros(len(haystack) + 16, dtype=np.uint8)
    hay[:len(haystack)] = np.frombuffer(haystack, dtype=np.uint8)
    cap = len(haystack) + 2
    out = np.zeros((cap, 3), dtype=np.uint64)
    cnt = C.c_uint64()
    f = lib().emu_ham_search_bits
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    rc = f(needle, n, hay.ctypes.data, len(haystack), k, tile, out.ctypes.data, cap, C.byref(cnt))
    if rc:
        return None
    return [(int(out[i, 0]), int(out[i, 1]), int(out[i, 2] & np.uint64(0xFFFFFFFF))) for i in range(cnt.value)]


def ham_search_phase(needle, haystack, k, tile=256, q_force=0):
    """All-mode hits of the phased bit-sliced hamming_search form (ham_phase_body.h) over a tiled haystack, with its plan and the number of
    candidates its filter passed: (hits, (Q, L, B), candidates); None where the form does not apply."""
    n = len(needle)
    hay = np.zeros(len(haystack) + 16, dtype=np.uint8)
    hay[:len(haystack)] = np.frombuffer(haystack, dtype=np.uint8)
    cap = len(haystack) + 2
    out = np.zeros((cap, 3), dtype=np.uint64)
    cnt, cand = C.c_uint64(), C.c_uint64()
    plan = (C.c_uint32 * 3)()
    f = lib().emu_ham_search_phase
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p,
                  C.c_void_p, C.c_void_p]
    rc = f(needle, n, hay.ctypes.data, len(haystack), k, tile, q_force, out.ctypes.data, cap, C.byref(cnt), plan, C.byref(cand))
    if rc:
        return None
    hits = [(int(out[i, 0]), int(out[i, 1]), int(out[i, 2] & np.uint64(0xFFFFFFFF))) for i in range(cnt.value)]
    return hits, (int(plan[0]), int(plan[1]), int(plan[2])), int(cand.value)


def ham_search_swar(needle, haystack, k, delta=0):
    """All-mode hits [(start, end, k)] of the SWAR hamming_search form (ham_swar_body.h: 16 offsets per lane) with the haystack starting
    `delta` bytes behind a 16-byte boundary."""
    n = len(needle)
    hay = np.zeros(len(haystack) + 16, dtype=np.uint8)
    hay[:len(haystack)] = np.frombuffer(haystack, dtype=np.uint8)
    cap = len(haystack) + 2
    out = np.zeros((cap, 3), dtype=np.uint64)
    cnt = C.c_uint64()
    f = lib().emu_ham_search_swar
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    rc = f(needle, n, hay.ctypes.data, len(haystack), k, delta, out.ctypes.data, cap, C.byref(cnt))
    if rc:
        raise RuntimeError("emu_ham_search_swar rc=%d" % rc)
    return [(int(out[i, 0]), int(out[i, 1]), int(out[i, 2] & np.uint64(0xFFFFFFFF))) for i in range(cnt.value)]


def lev_widebits_huge(a, b, k, trans=False, nwl=1, tile_steps=256, order=0):
    """One pair through the tiled (many-wavefront) form of the row-blocked kernel; order = tile order inside a launch."""
    a = bytes(a) + b"\0" * 16
    b = bytes(b) + b"\0" * 16
    v = lib().emu_lev_widebits_huge(a, len(a) - 16, b, len(b) - 16, k, int(bool(trans)), nwl, tile_steps, order)
    return None if v == 0xFFFFFFFF else int(v)


_EDIT_NAMES = ["Match", "Mismatch", "AGap", "BGap", "Transpose"]


def lev_widebits_trace(a, b, k, trans=False, nwl=1):
    """(distance | None, run-length edits | None) of the TRACE form of the row-blocked kernel + the host walk."""
    a2 = bytes(a) + b"\0" * 16
    b2 = bytes(b) + b"\0" * 16
    cap = len(a) + len(b) + 4
    out = np.zeros(cap, dtype=[("code", np.uint32), ("pad", np.uint32), ("count", np.uint64)])
    dist = C.c_uint32()
    n = C.c_uint64()
    rc = lib().emu_lev_widebits_trace(a2, len(a), b2, len(b), k, int(bool(trans)), nwl, C.byref(dist), out.ctypes.data, cap, C.byref(n))
    if rc:
        raise RuntimeError("emu_lev_widebits_trace rc=%d" % rc)
    if dist.value == 0xFFFFFFFF:
        return None, None
    return int(dist.value), [(_EDIT_NAMES[int(out[i]["code"])], int(out[i]["count"])) for i in range(n.value)]


def sliced_plan(a_len, b_len, unit_k):
    """lev_plan.h lev_sliced_make_plan -> dict (ok, S, dhi, c_ans, e_ans, dabs, steps)."""
    import numpy as np
    out = np.zeros(7, dtype=np.int64)
    lib().emu_sliced_plan(a_len, b_len, unit_k, out.ctypes.data)
    return dict(zip(("ok", "S", "dhi", "c_ans", "e_ans", "dabs", "steps"), (int(v) for v in out)))
