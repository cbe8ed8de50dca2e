"""Times the band-wavefront kernel for several lane layouts (TA_FORCE_D / TA_FORCE_L) on one workload."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import oracle_lib as O
import triple_accel_amd as T
from triple_accel_amd import batch as B

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n, L, k, costs = {"cfg2": (1_000_000, 256, 32, (1, 1, 0, None)), "cfg4": (1_000_000, 128, 8, (1, 1, 0, 1))}[wl]
a, b = Dg.pairs_random(1, n, L)
sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
out = torch.empty(n, dtype=torch.int32, device="cuda")
cells = O.band_cells(L, L, k, costs)
want = O.levenshtein_k_batch(O.csr_from_fixed(a[:2000]), O.csr_from_fixed(b[:2000]), k, costs)
layouts = [tuple(map(int, x.split(","))) for x in sys.argv[2:]] or [(0, 0)]
for D, Lp in layouts:
    if D or Lp:
        os.environ["TA_FORCE_D"] = str(D); os.environ["TA_FORCE_L"] = str(Lp)
    else:
        os.environ.pop("TA_FORCE_D", None); os.environ.pop("TA_FORCE_L", None)
    try:
        B.levenshtein_k_batch(sa, sb, k, costs, out=out)
    except Exception as e:
        print(D, Lp, "ERR", e); continue
    torch.cuda.synchronize()
    ok = np.array_equal(out[:2000].cpu().numpy().view(np.uint32), want)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        B.levenshtein_k_batch(sa, sb, k, costs, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    info = T.last_launch_info()
    print("kernel=%d D=%d L=%d PW=%d  %.3f ms  %.0f GCUPS  parity=%s" % (info["kernel"], info["diags_per_lane"], info["lanes_per_pair"],
          info["pairs_per_wave"], ms, cells * n / ms / 1e6, ok), flush=True)
