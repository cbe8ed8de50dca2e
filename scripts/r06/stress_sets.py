"""Device-set churn on a GPU box: thousands of pool switches with tiny jobs in between (round 6's fuzz run died of heap corruption after ~7,500 switches).
usage: python scripts/r06/stress_sets.py <iterations> <mode>   mode: jobs | nojobs | search"""
import os, sys, time
os.environ.setdefault("TA_TUNING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import triple_accel_amd as T
from triple_accel_amd import multi as M
n_it, mode = int(sys.argv[1]), sys.argv[2]
os.environ["TA_MULTI_MIN_PAIRS"] = "1"
os.environ["TA_MULTI_MIN_HAY"] = "8"
a, b = [b"kitten", b"abc", b"", b"flaw"] * 8, [b"sitting", b"abd", b"xy", b"lawn"] * 8
hay = b"xxabcdefghijklmnopyy" * 40
t0 = time.time()
for it in range(n_it):
    M.set_devices([0] * (1 + it % 8))
    if mode == "jobs":
        assert M.levenshtein_k_batch_host(a, b, 3).tolist() == [3, 1, 2, 2] * 8
    elif mode == "search":
        assert len(list(T.levenshtein_search_simd_with_opts(b"abcdefghijklmnop", hay, 2, T.SearchType.All, T.LEVENSHTEIN_COSTS, False))) > 0
    if it % 1000 == 0:
        print(it, round(time.time() - t0, 1), flush=True)
print("ok", n_it, mode, flush=True)
