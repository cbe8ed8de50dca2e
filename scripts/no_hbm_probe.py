"""How much of a pass is memory: the same batch geometry with every pair reading the SAME two strings (stride 0: all loads hit the
caches) against the real batch.  usage: python scripts/no_hbm_probe.py   (GPU box; prints device ms per pass, median of 30)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import triple_accel_amd as T
from triple_accel_amd import batch as B

def ms(fn, reps=30, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

g = np.random.default_rng(5)
CASES = [("cfg2", 1_000_000, 256, 32, (1, 1, 0, None)), ("cfg4", 1_000_000, 128, 8, (1, 1, 0, 1)),
         ("cfg2w", 1_000_000, 256, 32, (2, 3, 1, None)), ("cfg4w", 1_000_000, 128, 8, (2, 2, 1, 3))]
for name, n, L, k, costs in CASES:
    a = g.integers(97, 123, size=(n, L), dtype=np.uint8); b = a.copy()
    pos = g.integers(0, L, size=(n, 3)); b[np.arange(n)[:, None], pos] = g.integers(97, 123, size=(n, 3), dtype=np.uint8)
    A, Bs = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    c = T.EditCosts(*costs)
    real = ms(lambda: B.levenshtein_k_batch(A, Bs, k, c, out=out))
    kern = T.last_kernel_name()
    A0 = B.Strings(A.blob, None, stride=0, length=L, n=n); B0 = B.Strings(Bs.blob, None, stride=0, length=L, n=n)
    same = ms(lambda: B.levenshtein_k_batch(A0, B0, k, c, out=out))
    print(f"{name}: real batch {real:.4f} ms, every pair the same two strings (no HBM reads) {same:.4f} ms   [{kern}]", flush=True)
