"""End-to-end (PCIe-inclusive) rate of the batch path and single-call latency, for DESIGN.md."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import triple_accel_amd as T
from triple_accel_amd import batch as B

n, L, k = 1_000_000, 256, 32
a, b = Dg.pairs_random(1, n, L)
ta = torch.from_numpy(a).pin_memory(); tb = torch.from_numpy(b).pin_memory()
da = torch.zeros(n * L + 16, dtype=torch.uint8, device="cuda"); db = torch.zeros_like(da)
out = torch.empty(n, dtype=torch.int32, device="cuda"); hout = torch.empty(n, dtype=torch.int32).pin_memory()
def e2e():
    da[: n * L].copy_(ta.reshape(-1), non_blocking=True); db[: n * L].copy_(tb.reshape(-1), non_blocking=True)
    B.levenshtein_k_batch(B.Strings(da, None, L, L, n=n), B.Strings(db, None, L, L, n=n), k, out=out)
    hout.copy_(out, non_blocking=True); torch.cuda.synchronize()
for _ in range(2): e2e()
t = time.perf_counter()
for _ in range(5): e2e()
dt = (time.perf_counter() - t) / 5
print("cfg2 end-to-end (pinned H2D 512 MB + kernel + D2H 4 MB): %.2f ms -> %.0f GCUPS" % (dt * 1e3, 15584 * n / dt / 1e9))
x, y = a[0].tobytes(), b[0].tobytes()
for _ in range(20): T.levenshtein_simd_k(x, y, 32)
t = time.perf_counter()
for _ in range(500): T.levenshtein_simd_k(x, y, 32)
print("single-call levenshtein_simd_k (256 B host buffers): %.1f us per call" % ((time.perf_counter() - t) / 500 * 1e6))
t = time.perf_counter()
for _ in range(500): T.hamming(x, y)
print("single-call hamming (256 B host buffers): %.1f us per call" % ((time.perf_counter() - t) / 500 * 1e6))
# FETCH_SIZE calibration workload: hamming over 1M x 256 B = 512 MB streamed with 16 B per lane
sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
for _ in range(3): B.hamming_batch(sa, sb, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): B.hamming_batch(sa, sb, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("hamming batch 1M x 256 B: %.3f ms -> %.0f GB/s" % (ms, (2 * L + 4) * n / ms / 1e6))
for _ in range(3): B.levenshtein_k_batch(sa, sb, k, out=out)
torch.cuda.synchronize()
