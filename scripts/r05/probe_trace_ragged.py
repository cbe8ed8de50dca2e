"""Timing probe (round 5, late): ta_levenshtein_trace_batch on a CSR batch -- 1M mutated pairs, lengths uniform on 32..256, k = 32 -- with the
pairs taken in length order (the default for CSR batches of >= 4,096 pairs) and in batch order (TA_NO_LENGTH_ORDER=1, needs TA_TUNING=1)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import triple_accel_amd as T
from triple_accel_amd import batch as B
n = int(os.environ.get("PROBE_PAIRS", "1000000"))
g = np.random.default_rng(7)
lens = g.integers(32, 257, n)
am, bm = Dg.pairs_mutated_fixed(5, n, 256, int(os.environ.get("PROBE_EDITS", "16")))
# ragged: pair i keeps the first lens[i] bytes of both strings (CSR blobs)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
idx = np.arange(256)[None, :] < lens[:, None]
blob_a, blob_b = am[idx], bm[idx]
sa = B.Strings.from_csr(blob_a, off) if hasattr(B.Strings, "from_csr") else B.Strings.from_list([am[i, :lens[i]].tobytes() for i in range(n)])
sb = B.Strings.from_csr(blob_b, off) if hasattr(B.Strings, "from_csr") else B.Strings.from_list([bm[i, :lens[i]].tobytes() for i in range(n)])
out = torch.empty(n, dtype=torch.int32, device="cuda"); ed = torch.empty((n, 65, 2), dtype=torch.int64, device="cuda"); ne = torch.empty(n, dtype=torch.int32, device="cuda")
for it in range(3):
    B.levenshtein_trace_batch(sa, sb, 32, cap=65, out=out, edits=ed, n_edits=ne)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for it in range(10):
    B.levenshtein_trace_batch(sa, sb, 32, cap=65, out=out, edits=ed, n_edits=ne)
ev1.record(); torch.cuda.synchronize()
print("ragged trace batch, batch order" if os.environ.get("TA_NO_LENGTH_ORDER") else "ragged trace batch, length order", "ms per pass", round(ev0.elapsed_time(ev1) / 10, 4), T.last_kernel_name(),
      "scripts", int((ne > 0).sum()))
