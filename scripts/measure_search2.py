"""Search timings for a few needle lengths over a 256 MiB shard (filter path vs exact kernel over everything)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
from triple_accel_amd import batch as B
g = Dg.rng(1)
mib = 256
hay_np = Dg.random_bytes(g, mib << 20)
hay = B.haystack_tensor(hay_np)
for n in (8, 16, 32, 48, 64, 128, 256):
    needle = Dg.random_bytes(Dg.rng(n), n).tobytes()
    k = n // 2
    for mode in ("filter", "exact"):
        if mode == "exact":
            if n > 64: continue
            os.environ["TA_SEARCH_NOFILTER"] = "1"
        else:
            os.environ.pop("TA_SEARCH_NOFILTER", None)
        B.levenshtein_search_dev(needle, hay, k); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): hits = B.levenshtein_search_dev(needle, hay, k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("n=%3d k=%3d %-6s %8.3f ms  %7.1f GCUPS  hits=%d" % (n, k, mode, dt * 1e3, n * (mib << 20) / dt / 1e9, len(hits)), flush=True)
