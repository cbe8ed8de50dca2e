"""Timing probe (round 5): the checkpoint trace kernel with and without its walk (TA_TRACE_SKIP_WALK=1: no scripts, the recomputation only)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import triple_accel_amd as T
from triple_accel_amd import batch as B
n = 1_000_000
E = int(os.environ.get("PROBE_EDITS", "32"))
am, bm = Dg.pairs_mutated_fixed(5, n, 256, E)
sa, sb = B.Strings.from_fixed(am), B.Strings.from_fixed(bm)
out = torch.empty(n, dtype=torch.int32, device="cuda"); ed = torch.empty((n, 65, 2), dtype=torch.int64, device="cuda"); ne = torch.empty(n, dtype=torch.int32, device="cuda")
for it in range(3):
    B.levenshtein_trace_batch(sa, sb, 32, cap=65, out=out, edits=ed, n_edits=ne)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for it in range(10):
    B.levenshtein_trace_batch(sa, sb, 32, cap=65, out=out, edits=ed, n_edits=ne)
ev1.record(); torch.cuda.synchronize()
print("edits", E, "skip_walk", os.environ.get("TA_TRACE_SKIP_WALK", "0"), "ms per pass", ev0.elapsed_time(ev1) / 10, T.last_kernel_name())
