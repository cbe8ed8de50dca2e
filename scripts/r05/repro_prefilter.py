import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg, oracle_lib as O
import triple_accel_amd as T
from triple_accel_amd import batch as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
am, bm = Dg.pairs_mutated_fixed(5, n, 256, 10)
sa, sb = B.Strings.from_fixed(am), B.Strings.from_fixed(bm)
T.set_option(T.OPT_UNIT_PREFILTER, True)
for it in range(3):
    out = B.levenshtein_k_batch(sa, sb, 32, (2, 3, 1, None))
    torch.cuda.synchronize()
    print("pass", it, T.last_kernel_name(), T.last_launch_info(), flush=True)
got = out.cpu().numpy().view(np.uint32)
T.set_option(T.OPT_UNIT_PREFILTER, False)
ref = B.levenshtein_k_batch(sa, sb, 32, (2, 3, 1, None)).cpu().numpy().view(np.uint32)
print("equal to the plain pass:", np.array_equal(got, ref), (got != 0xFFFFFFFF).sum())
