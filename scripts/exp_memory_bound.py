"""How much of the band kernel's time is memory?  The same launch (cfg2: 1M pairs, 256 B, k = 32; EXP_WL=cfg4: 128 B, k = 8, RDAMERAU)
with every pair reading the SAME bytes (stride 0: all hits in L1/L2, no HBM traffic) against the real batch; and the real batch
under the environment switches given as arguments ("TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=40000" ...; needs TA_TUNING=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import datagen as Dg
import triple_accel_amd as T
from triple_accel_amd import batch as B

CFG4 = os.environ.get("EXP_WL") == "cfg4"
n, L, k = (1_000_000, 128, 8) if CFG4 else (1_000_000, 256, 32)
COSTS = T.RDAMERAU_COSTS if CFG4 else T.LEVENSHTEIN_COSTS
a, b = Dg.pairs_random(0x7A04 if CFG4 else 0x7A02, n, L)
sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
out = torch.empty(n, dtype=torch.int32, device="cuda")

def dev_ms(fa, fb, reps=50):
    for _ in range(5): B.levenshtein_k_batch(fa, fb, k, COSTS, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): B.levenshtein_k_batch(fa, fb, k, COSTS, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

t_end = time.time() + 0.5                      # clocks come back from idle over tens of passes: ramp before the first figure
while time.time() < t_end:
    for _ in range(50): B.levenshtein_k_batch(sa, sb, k, COSTS, out=out)
    torch.cuda.synchronize()
s0a = B.Strings(sa.blob, None, stride=0, length=L, n=n)
s0b = B.Strings(sb.blob, None, stride=0, length=L, n=n)
print("real batch           %.4f ms" % dev_ms(sa, sb), T.last_launch_info())
print("stride 0 (no HBM)    %.4f ms" % dev_ms(s0a, s0b))
# EXP_STRIDES=16,128,...: pair i reads its strings at byte i * stride of the same blobs (overlapping strings: a smaller footprint --
# 16 B: 16 MB per string set, L2-resident, 8 lines per load instruction; 128 B: 128 MB, one line per lane as in the real batch)
for st in [int(x) for x in os.environ.get("EXP_STRIDES", "").split(",") if x]:
    xa = B.Strings(sa.blob, None, stride=st, length=L, n=n)
    xb = B.Strings(sb.blob, None, stride=st, length=L, n=n)
    print("stride %-5d         %.4f ms" % (st, dev_ms(xa, xb)))
print("real batch again     %.4f ms" % dev_ms(sa, sb))
for env in sys.argv[1:]:
    kv = dict(x.split("=") for x in env.split(",")) if env else {}
    os.environ.update(kv)
    print("%-40s %.4f ms  lds %d grid %d" % (env, dev_ms(sa, sb), T.last_launch_info()["lds_bytes"], T.last_launch_info()["grid"]))
    for key in kv: os.environ.pop(key)
