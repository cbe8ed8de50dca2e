import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import triple_accel_amd as T
g = Dg.rng(1)
import numpy as np
a = g.integers(33, 127, size=(5_000_000, 64), dtype=np.uint8)
x = Dg.rand_str(g, 200_000); y = Dg.mutate(g, x, 3000, True)
T.levenshtein(x[:1000], y[:1000])
for k in (30, 120, 240, 960, 1920, 3840, 0xFFFFFFFF):
    t = time.perf_counter(); d = T.levenshtein_simd_k(x, y, k); dt = time.perf_counter() - t
    print("k=%d -> %s in %.1f ms, kernel %d" % (k, d, dt * 1e3, T.last_launch_info()["kernel"]), flush=True)
