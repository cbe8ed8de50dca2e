import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
from triple_accel_amd import batch as B
g = Dg.rng(1)
hay_np = Dg.random_bytes(g, 1 << 30)
needle = Dg.random_bytes(g, 32).tobytes()
for pos in range(1 << 16, hay_np.size - 100, 1 << 20):
    hay_np[pos:pos + 32] = np.frombuffer(needle, dtype=np.uint8)
    hay_np[pos + 5] = 7
hay = B.haystack_tensor(hay_np)
for n in (8, 32, 128):
    nd = needle[:n] if n <= 32 else (needle * 4)[:n]
    hits = B.hamming_search_dev(nd, hay, n // 4)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): hits = B.hamming_search_dev(nd, hay, n // 4)
    dt = (time.perf_counter() - t) / 3
    print("hamming_search needle %d over 1 GiB: %.2f ms, %d hits, %.1f G positions/s, %.0f GB/s algorithmic" % (n, dt * 1e3, len(hits), (1 << 30) / dt / 1e9, (1 << 30) / dt / 1e9))
