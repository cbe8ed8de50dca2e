import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import triple_accel_amd as T
from triple_accel_amd import batch as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
am, bm = Dg.pairs_mutated_fixed(5, n, 256, 10)
sa, sb = B.Strings.from_fixed(am), B.Strings.from_fixed(bm)
ref = B.levenshtein_k_batch(sa, sb, 32, (2, 3, 1, None)).cpu().numpy().view(np.uint32)
T.set_option(T.OPT_UNIT_PREFILTER, True)
if len(sys.argv) > 3:                                      # a small batch through the same path first (the bench's side batch)
    a2, b2 = Dg.pairs_mutated_fixed(6, 2048, 256, 10)
    side_out = torch.empty(2048, dtype=torch.int32, device="cuda")
    B.levenshtein_k_batch(B.Strings.from_fixed(a2), B.Strings.from_fixed(b2), 32, (2, 3, 1, None), out=side_out)
    torch.cuda.synchronize()
    del side_out
out = torch.empty(n, dtype=torch.int32, device="cuda")
B.levenshtein_k_batch(sa, sb, 32, (2, 3, 1, None), out=out)
torch.cuda.synchronize()
print("eager ok", np.array_equal(out.cpu().numpy().view(np.uint32), ref), flush=True)
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
        B.levenshtein_k_batch(sa, sb, 32, (2, 3, 1, None), out=out)
torch.cuda.synchronize()
print("captured", flush=True)
out.fill_(5)
g.replay()
torch.cuda.synchronize()
print("replayed", np.array_equal(out.cpu().numpy().view(np.uint32), ref), flush=True)
for it in range(30):
    B.levenshtein_k_batch(sa, sb, 32, (2, 3, 1, None), out=out)
torch.cuda.synchronize()
print("eager after capture ok", flush=True)
for it in range(3):
    g.replay()
torch.cuda.synchronize()
print("replayed again", np.array_equal(out.cpu().numpy().view(np.uint32), ref), flush=True)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); g.replay(); ev1.record(); torch.cuda.synchronize()
print("timed", ev0.elapsed_time(ev1), flush=True)
