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
