"""Single-call latency of the host API (host buffers in, result out), microseconds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import triple_accel_amd as T

g = Dg.rng(3)
def lat(f, reps=300):
    for _ in range(20): f()
    t = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t) / reps * 1e6

for n in (16, 256, 4096):
    x = Dg.rand_str(g, n); y = Dg.mutate(g, x, max(1, n // 20))
    print("len %5d: hamming %6.1f  simd_k(k=8) %6.1f  simd_k(k=32) %6.1f  levenshtein %6.1f  exp %6.1f  rdamerau %6.1f  search(k=2) %6.1f us" % (
        n, lat(lambda: T.hamming(x, x)), lat(lambda: T.levenshtein_simd_k(x, y, 8)), lat(lambda: T.levenshtein_simd_k(x, y, 32)),
        lat(lambda: T.levenshtein(x, y)), lat(lambda: T.levenshtein_exp(x, y)), lat(lambda: T.rdamerau(x, y)),
        lat(lambda: T.levenshtein_search_simd_with_opts(x[:8], y, 2, T.SearchType.Best, T.LEVENSHTEIN_COSTS, False))), flush=True)

# the queue: pairs pushed one at a time, one batch pass per flush (amortised cost per pair, host buffers in, answers out)
for n in (256,):
    x = Dg.rand_str(g, n); y = Dg.mutate(g, x, max(1, n // 20))
    for per_flush in (100, 1000, 10000):
        q = T.Queue(32)
        def round_trip():
            for _ in range(per_flush): q.push(x, y)
            return q.flush()
        us = lat(round_trip, reps=20) / per_flush
        print("queue, %d-byte pairs, k = 32, %5d pairs per flush: %6.2f us per pair (push + its share of the flush)" % (n, per_flush, us), flush=True)
        q.close()
