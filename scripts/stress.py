"""A few out-of-the-ordinary shapes (not part of the test suite): many tiny pairs, one huge pair, a big haystack."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import oracle_lib as O
import triple_accel_amd as T
from triple_accel_amd import batch as B

g = Dg.rng(1)
# (a) 5M pairs x 64 B, k = 5
a = g.integers(33, 127, size=(5_000_000, 64), dtype=np.uint8); b = a.copy()
pos = g.integers(0, 64, size=(a.shape[0], 3)); b[np.arange(a.shape[0])[:, None], pos] = 32
sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
t = time.perf_counter(); out = B.levenshtein_k_batch(sa, sb, 5).cpu().numpy().view(np.uint32); dt = time.perf_counter() - t
idx = np.r_[0:1000, -1000:0]
assert np.array_equal(out[idx], O.levenshtein_k_batch(O.csr_from_fixed(a[idx]), O.csr_from_fixed(b[idx]), 5))
print("5M x 64 B: %.1f ms, kernel %d" % (dt * 1e3, T.last_launch_info()["kernel"]), flush=True)
# (b) one huge pair
x = Dg.rand_str(g, 200_000); y = Dg.mutate(g, x, 3000, True)
t = time.perf_counter(); d = T.levenshtein(x, y); dt = time.perf_counter() - t
print("200K x 200K levenshtein: d = %d in %.2f s, kernel %d" % (d, dt, T.last_launch_info()["kernel"]), flush=True)
t = time.perf_counter(); d2 = T.levenshtein_exp(x, y); dt = time.perf_counter() - t
print("200K x 200K levenshtein_exp: d = %d in %.2f s" % (d2, dt), flush=True)
assert d == d2
t = time.perf_counter(); want = O.levenshtein_exp(x, y); print("oracle exp: %d in %.1f s" % (want, time.perf_counter() - t)); assert want == d
# (c) 3 GiB haystack
hay = torch.randint(1, 256, ((3 << 30) + 16,), dtype=torch.uint8, device="cuda")
needle = bytes(range(40, 72))
hay[1_000_000:1_000_032] = torch.tensor(list(needle), dtype=torch.uint8, device="cuda")
hay[(3 << 30) - 32:(3 << 30)] = torch.tensor(list(needle), dtype=torch.uint8, device="cuda")
t = time.perf_counter(); hits = B.levenshtein_search_dev(needle, (hay, 3 << 30), 8); dt = time.perf_counter() - t
print("3 GiB search: %.1f ms, %d hits, first %s last %s" % (dt * 1e3, len(hits), hits[0], hits[-1]), flush=True)
assert any(h[1] == 1_000_032 and h[2] == 0 for h in hits) and any(h[1] == (3 << 30) and h[2] == 0 for h in hits)
h2 = B.hamming_search_dev(needle, (hay, 3 << 30), 2)
print("3 GiB hamming_search: %d hits" % len(h2), flush=True)
assert any(h[0] == 1_000_000 for h in h2)
print("stress ok")
