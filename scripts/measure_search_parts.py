"""cfg5 search pass, piece by piece: the C call (filter + exact kernel + count read-back), device sort + copy, host fold."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, ctypes as C
import datagen as Dg
import triple_accel_amd as T
from triple_accel_amd import batch as B, dist as TD, _native as N

g = Dg.rng(0x7A05 + 5)
needle = Dg.random_bytes(Dg.rng(0x7A05), 32).tobytes()
hay_np = Dg.random_bytes(g, 1 << 30)
for pos in range(1 << 16, hay_np.size - 100, 1 << 20):
    mm = np.frombuffer(Dg.mutate(g, needle, 10), dtype=np.uint8); hay_np[pos:pos + mm.size] = mm
hay = B.haystack_tensor(hay_np)
def t(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
ms_all, hits = t(lambda: B.levenshtein_search_dev(needle, hay, 16, T.LEVENSHTEIN_COSTS))
print("search_dev (C call + sort + copy): %.3f ms, %d hits" % (ms_all, len(hits)))
h, length = hay
cap = min(length + 2, 1 << 22)
buf = B._hit_buffer(h.device, cap); count = C.c_uint64(); cc = T.LEVENSHTEIN_COSTS._c() if hasattr(T.LEVENSHTEIN_COSTS, "_c") else B._costs(T.LEVENSHTEIN_COSTS)._c()
def ccall():
    rc = N.lib().ta_levenshtein_search_dev(needle, len(needle), h.data_ptr(), length, 16, C.byref(cc), 0, 0, 0, buf.data_ptr(), cap, C.byref(count), B._stream())
    assert rc == 0
    return int(count.value)
ms_c, n = t(ccall)
print("C call alone: %.3f ms (%d hits)" % (ms_c, n))
ms_s, rows = t(lambda: B._hits_to_numpy(buf, n))
print("device sort + copy: %.3f ms" % ms_s)
ms_f, best = t(lambda: TD.fold_best(rows, 16, True))
print("host fold: %.3f ms -> %d matches" % (ms_f, len(best)))
ms_b, rows_b = t(lambda: B.levenshtein_search_best_dev(needle, hay, 16, T.LEVENSHTEIN_COSTS))
print("search_best_dev (C call + on-device selection): %.3f ms, %d rows" % (ms_b, len(rows_b)))
ms_a, _ = t(lambda: TD.fold_best(B.levenshtein_search_dev(needle, hay, 16, T.LEVENSHTEIN_COSTS), 16, True))
ms_n, _ = t(lambda: TD.fold_best(B.levenshtein_search_best_dev(needle, hay, 16, T.LEVENSHTEIN_COSTS), 16, True))
print("whole pass: all hits to the host %.3f ms, best-k hits only %.3f ms" % (ms_a, ms_n))
# the exact kernel's own clock (100 MHz timestamps in the report of the last pass)
rep = (C.c_uint32 * 16)()
N.lib().ta_debug_last_search_report.argtypes = [C.c_void_p]
for name, f in (("best", lambda: B.levenshtein_search_best_dev(needle, hay, 16, T.LEVENSHTEIN_COSTS)), ("all", ccall)):
    for _ in range(3):
        f(); torch.cuda.synchronize()
        N.lib().ta_debug_last_search_report(rep)
        tt = [int(rep[8 + i]) for i in range(4)]
        d = lambda a, b: ((b - a) & 0xFFFFFFFF) / 100.0
        print("%s pass: count %d  n_list %d  dense %d  sel %d (state %d)  min_k %d  n_cand %d | wavefront 0: %.1f us in its blocks; report starts "
              "%.1f us after wavefront 0 entered, takes %.1f us" % (name, rep[0] | (rep[1] << 32), rep[2], rep[3], rep[4], rep[5], rep[6], rep[7],
                                                                   d(tt[0], tt[1]), d(tt[0], tt[2]), d(tt[2], tt[3])))
