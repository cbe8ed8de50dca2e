"""Prototype of the row-blocked bit-parallel column step used by lev_widebits (blocks of RB rows, horizontal
differences handed from block to block, Hyyro's transposition term across block edges)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def blocks_distance(a, b, trans, RB, variant):
    n, m = len(a), len(b)
    nb = max(1, (n + RB - 1) // RB)
    mask = (1 << RB) - 1
    top = 1 << (RB - 1)
    Pv = [mask] * nb; Mv = [0] * nb
    D0p = [mask] * nb; Eqp = [0] * nb
    for j in range(1, m + 1):
        rP, rM, rX = 1, 0, 0          # row 0: D[0][j] - D[0][j-1] = +1
        for t in range(nb):
            Eq = 0
            for r in range(RB):
                i = t * RB + r        # 0-based row of a
                if i < n and a[i] == b[j - 1]:
                    Eq |= 1 << r
            hN = rM
            Eq1 = Eq | hN
            s = ((Eq1 & Pv[t]) + Pv[t]) & mask
            D0 = ((s ^ Pv[t]) | Eq1 | Mv[t]) & mask
            X = (~D0p[t] & mask) & Eq
            if trans:
                TR = (((X << 1) | rX) & mask) & Eqp[t]
                D0 |= TR
            Ph = (Mv[t] | ~(D0 | Pv[t])) & mask
            Mh = D0 & Pv[t]
            oP, oM, oX = (Ph >> (RB - 1)) & 1, (Mh >> (RB - 1)) & 1, (X >> (RB - 1)) & 1
            Phs = ((Ph << 1) | rP) & mask
            Mhs = ((Mh << 1) | rM) & mask
            if variant == "hyyro":
                Xv = D0
            else:
                Xv = (Eq | Mv[t]) if not trans else (D0 & ~hN | (Eq | Mv[t]) & hN)
            Pv[t] = (Mhs | ~(Xv | Phs)) & mask
            Mv[t] = Phs & Xv
            D0p[t], Eqp[t] = D0, Eq
            rP, rM, rX = oP, oM, oX
    # D[n][m] = m + sum of vertical differences of column m over rows 1..n
    d = m
    for t in range(nb):
        for r in range(RB):
            if t * RB + r < n:
                d += ((Pv[t] >> r) & 1) - ((Mv[t] >> r) & 1)
    return d


if __name__ == "__main__":
    import numpy as np
    import oracle_lib as O
    g = np.random.default_rng(3)
    for variant in ("hyyro", "myers"):
        bad = tot = 0
        for it in range(3000):
            A = int(g.integers(2, 6))
            x = bytes(g.integers(97, 97 + A, size=int(g.integers(0, 40))).astype(np.uint8))
            y = bytes(g.integers(97, 97 + A, size=int(g.integers(0, 40))).astype(np.uint8))
            for trans in (False, True):
                want = O.rdamerau(x, y) if trans else O.levenshtein(x, y)
                for RB in (1, 2, 3, 4, 8, 64):
                    got = blocks_distance(x, y, trans, RB, variant)
                    tot += 1
                    if got != want:
                        bad += 1
                        if bad < 4: print(variant, "MISMATCH", x, y, trans, RB, got, want)
        print(variant, "checked", tot, "bad", bad)
