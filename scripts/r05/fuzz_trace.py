"""Randomised parity run of ta_levenshtein_trace_batch's checkpoint-and-recompute route (round 5, late: after the trace kernel's strings moved
into registers): fixed-length batches (the forward sweep folded into the distance pass), CSR batches (the kernel's own sweep, every tile form),
both unit-cost families, against the oracle's scripts edit for edit.  usage: python scripts/r05/fuzz_trace.py <minutes> [seed]  (GPU box)"""
import os, sys, time
os.environ.setdefault("TA_TUNING", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import oracle_lib as O
import triple_accel_amd as T
from triple_accel_amd import batch as B

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
g = np.random.default_rng(seed)
print("seed", seed, flush=True)
t_end, rounds, pairs, somes, kinds = time.time() + 60 * minutes, 0, 0, 0, {}
while time.time() < t_end:
    rounds += 1
    for sw in ("TA_TRACE_TILE", "TA_TRACE_STILE", "TA_TRACE_OWN_SWEEP", "TA_TRACE_CSR_OWN_SWEEP"):
        os.environ.pop(sw, None)
    if g.random() < 0.3:
        os.environ["TA_TRACE_TILE"] = str(g.choice([8, 16, 32]))
    if g.random() < 0.5:
        os.environ["TA_TRACE_STILE"] = str(g.choice([32, 64]))
    if g.random() < 0.2:
        os.environ["TA_TRACE_OWN_SWEEP"] = "1"
    if g.random() < 0.2:
        os.environ["TA_TRACE_CSR_OWN_SWEEP"] = "1"
    alpha = [(1, 256), (97, 101), (0, 256), (12, 14), (33, 127)][int(g.integers(0, 5))]
    costs = [(1, 1, 0, None), (1, 1, 0, 1)][int(g.integers(0, 2))]
    k = int(g.choice([0, 1, 3, 10, 20, 29, 30, 32]))
    n = int(g.choice([1, 40, 64, 65, 700, 3000, 5000]))            # (CSR batches of >= 4,096 pairs: length order)
    fixed = bool(g.random() < 0.6)
    if fixed:
        la = int(g.choice([1, 2, 7, 15, 16, 17, 63, 64, 65, 100, 128, 255, 256, 257, 500, 1100]))
        lb = la if g.random() < 0.6 else max(1, la + int(g.integers(-12, 13)))
        r = Dg.rng(int(g.integers(1 << 30)))
        fa = g.integers(alpha[0], alpha[1], (n, la), dtype=np.uint8)
        fb = np.empty((n, lb), dtype=np.uint8)
        for i in range(n):
            if g.random() < 0.85 and la:
                y = Dg.mutate(r, fa[i].tobytes(), int(g.integers(0, 1 + int(g.choice([2, 10, 30])))), True)
                y = (y + bytes(g.integers(alpha[0], alpha[1], lb, dtype=np.uint8)))[:lb]
            else:
                y = g.integers(alpha[0], alpha[1], lb, dtype=np.uint8).tobytes()
            fb[i] = np.frombuffer(y, dtype=np.uint8)
        a = [fa[i].tobytes() for i in range(n)]; b = [fb[i].tobytes() for i in range(n)]
        sa, sb = B.Strings.from_fixed(fa), B.Strings.from_fixed(fb)
    else:
        hi = int(g.choice([12, 60, 200, 500, 1400]))
        lo = int(g.integers(0, hi + 1))
        a, b = [], []
        r = Dg.rng(int(g.integers(1 << 30)))
        for _ in range(n):
            x = g.integers(alpha[0], alpha[1], int(g.integers(lo, hi + 1)), dtype=np.uint8).tobytes()
            if g.random() < 0.8:
                y = Dg.mutate(r, x, int(g.integers(0, 1 + int(g.choice([2, 10, 30])))), True) if x else x
            else:
                y = g.integers(alpha[0], alpha[1], int(g.integers(lo, hi + 1)), dtype=np.uint8).tobytes()
            a.append(x); b.append(y)
        sa, sb = B.Strings.from_list(a), B.Strings.from_list(b)
    out, ed, ne = B.levenshtein_trace_batch(sa, sb, k, costs)
    name = T.last_kernel_name()
    gd = out.cpu().numpy().view(np.uint32); ge = B.edits_to_lists(ed, ne)
    step = max(1, n // 400)                      # (the oracle's scalar traceback is the slow side: a sample of big batches, every pair of small ones)
    for i in range(0, n, step):
        wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, costs)
        ok = (gd[i] == 0xFFFFFFFF and ge[i] == []) if wd is None else (gd[i] == wd and ge[i] == [tuple(e) for e in we])
        pairs += 1; somes += wd is not None
        if not ok:
            print("MISMATCH", dict(fixed=fixed, n=n, k=k, costs=costs, alpha=alpha, kernel=name, env={s: os.environ.get(s) for s in ("TA_TRACE_TILE", "TA_TRACE_STILE", "TA_TRACE_OWN_SWEEP", "TA_TRACE_CSR_OWN_SWEEP")}))
            print("pair", i, a[i], b[i], gd[i], ge[i], wd, we)
            sys.exit(1)
    kinds[name] = kinds.get(name, 0) + 1
print("rounds", rounds, "pairs checked", pairs, "with a script", somes, "no mismatch")
for kname in sorted(kinds):
    print("  ", kinds[kname], kname)
