"""Why does a 20-step timed region report a slower pass than a 100-step one?  Per-step HIP-event times of the driver's protocol
(sync, K back-to-back passes, sync), for K = 20 and 100, with and without work queued right up to the synchronisation point."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import datagen as Dg
from triple_accel_amd import batch as B

n, L, k = 1_000_000, 256, 32
a, b = Dg.pairs_random(0x7A02, n, L)
sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
out = torch.empty(n, dtype=torch.int32, device="cuda")
run = lambda: B.levenshtein_k_batch(sa, sb, k, out=out)

def region(steps, warm):
    for _ in range(warm): run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter(); ev[0].record()
    for i in range(steps):
        run(); ev[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    return wall, ms

for rep in range(2):
    time.sleep(3.0)                                   # the host-side set-up of a real run leaves the GPU idle for seconds
    for steps, warm in ((20, 5), (100, 5), (20, 800), (20, 5)):
        wall, ms = region(steps, warm)
        print("steps %3d warm %3d: wall %.4f ms/step; event ms: first3 %s  median %.4f  last %.4f  max %.4f" % (
            steps, warm, wall, " ".join("%.3f" % x for x in ms[:3]), float(np.median(ms)), ms[-1], max(ms)), flush=True)
