"""Round-4 probe H (GPU box): hamming_search -- shift-add scan against the SWAR kernel by needle length."""
import os, sys
os.environ["TA_TUNING"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
from triple_accel_amd import batch as B

def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

g = Dg.rng(1)
hay_np = Dg.random_bytes(g, 1 << 30); hay_np[hay_np == 0] = 1
needle = bytes(c or 1 for c in Dg.random_bytes(g, 64).tobytes())
for pos in range(1 << 16, hay_np.size - 100, 1 << 20):
    hay_np[pos:pos + 64] = np.frombuffer(needle, dtype=np.uint8); hay_np[pos + 5] = 7
hay = B.haystack_tensor(hay_np)
for n in (4, 8, 12, 16, 24, 32, 48, 64):
    row = []
    for env in ({}, {"TA_HAMMING_SEARCH_SWAR": "1"}):
        for k_, v_ in env.items(): os.environ[k_] = v_
        hits = B.hamming_search_dev(needle[:n], hay, max(1, n // 4))
        ms = t_ms(lambda: B.hamming_search_dev(needle[:n], hay, max(1, n // 4)))
        row.append((ms, len(hits)))
        for k_ in env: os.environ.pop(k_)
    print("needle %2d: default %.3f ms (%d hits) | SWAR %.3f ms (%d hits)  -> %.0f / %.0f GB/s" % (n, row[0][0], row[0][1], row[1][0], row[1][1], (1 << 30) / row[0][0] / 1e6, (1 << 30) / row[1][0] / 1e6), flush=True)
