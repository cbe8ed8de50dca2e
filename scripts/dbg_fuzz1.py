import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import datagen as Dg, oracle_lib as O
import triple_accel_amd as T
from triple_accel_amd import batch as B
os.environ["TA_DEBUG"] = "1"
print("device_count", T.device_count())
g = np.random.default_rng(1)
a = g.integers(97, 101, size=(500, 700), dtype=np.uint8); b = a.copy(); b[:, ::9] = 97
for costs in [(1, 1, 0, None), (5, 3, 0, 4)]:
    try:
        out = B.levenshtein_k_batch(B.Strings.from_fixed(a), B.Strings.from_fixed(b), 100, costs).cpu().numpy().view(np.uint32)
        print(costs, "ok", T.last_launch_info()["kernel"], np.array_equal(out, O.levenshtein_k_batch(O.csr_from_fixed(a), O.csr_from_fixed(b), 100, costs)))
    except Exception as e:
        print(costs, "EXC", e)
