"""Round-4 probe B (GPU box): where does the ragged CSR pass lose against the fixed-length one?  One variable at a time."""
import os, sys
os.environ["TA_TUNING"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import triple_accel_amd as T
from triple_accel_amd import batch as B

def t_ms(fn, reps=30):
    fn(); torch.cuda.synchronize()
    for _ in range(300): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

LEV = (1, 1, 0, None)
n, L, k = 1_000_000, 256, 32

def run_case(tag, la, lb, align=1, env=None):
    g = Dg.rng(11)
    for kk, vv in (env or {}).items(): os.environ[kk] = vv
    sides = []
    for ln in (la, lb):
        if align == 1:
            off = np.zeros(n + 1, dtype=np.int64); np.cumsum(ln, out=off[1:])
            blob = np.zeros(int(off[-1]) + 16, dtype=np.uint8); blob[:int(off[-1])] = Dg.random_bytes(g, int(off[-1]))
            sides.append(B.Strings(torch.from_numpy(blob).cuda(), torch.from_numpy(off).cuda(), max_len=int(ln.max())))
        else:
            # aligned starts: CSR offsets must be contiguous, so lengths are rounded up to `align` for the offsets and the
            # real content is followed by filler bytes that DIFFER between a and b (filler 1 vs 2) -- distances change, timing geometry
            # (length classes, band) is that of the rounded lengths.  Used only with lengths that are already multiples of align.
            raise SystemExit("aligned variants use lengths that are multiples of the alignment")
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    ms = t_ms(lambda: B.levenshtein_k_batch(sides[0], sides[1], k, LEV, out=out))
    cols = int(lb.sum())
    print("%-58s %.4f ms  %.3f ps/col  kernel %s" % (tag, ms, ms * 1e9 / cols, T.last_kernel_name()), flush=True)
    for kk in (env or {}): os.environ.pop(kk, None)

g = Dg.rng(3)
la = g.integers(32, L + 1, size=n).astype(np.int64)
lb = np.clip(la + g.integers(-4, 5, size=n), 1, L).astype(np.int64)
run_case("R0 ragged 32..256, b = a +- 4 (the bench batch)", la, lb)
run_case("R0a same, shortest first (TA_ORDER_ASC)", la, lb, env={"TA_ORDER_ASC": "1"})
run_case("R0v same, VLINE fetch form + exact length classes (TA_BITS_VLINE)", la, lb, env={"TA_BITS_VLINE": "1"})
run_case("R1 same, batch order (TA_NO_LENGTH_ORDER)", la, lb, env={"TA_NO_LENGTH_ORDER": "1"})
run_case("R4 ragged 32..256, b = a", la, la)
la8 = (la // 8) * 8
run_case("R5 lengths multiples of 8, b = a (starts 8-aligned)", la8, la8)
la16 = np.maximum((la // 16) * 16, 32)
run_case("R6 lengths multiples of 16, b = a (starts 16-aligned)", la16, la16)
la128 = np.where(la < 192, 128, 256)
run_case("R7 lengths 128 or 256, b = a (starts line-aligned)", la128, la128)
lc = np.full(n, 144, dtype=np.int64)
run_case("R8 all 144, b = a", lc, lc)
lc = np.full(n, 143, dtype=np.int64)
run_case("R9 all 143, b = a (unaligned starts)", lc, lc)
run_case("R9v all 143, VLINE fetch form", lc, lc, env={"TA_BITS_VLINE": "1"})
