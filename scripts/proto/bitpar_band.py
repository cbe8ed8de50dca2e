"""Prototype (Python ints) of the banded bit-parallel Levenshtein / restricted-Damerau step used by lev_bits_body.h.
Hyyro 2003, "A bit-vector algorithm for computing Levenshtein and Damerau edit distances", diagonal-band form:
the w-bit window slides one row down per column, so a diagonal is a fixed bit position."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def bitpar_band(a, b, k, trans):
    n, m = len(a), len(b)
    if n == 0 and m == 0:
        return 0
    K = min(k, max(n, m))          # unit costs: the distance never exceeds max(n, m)
    u = K                          # unit_k = (K - sg) / gc with sg = 0, gc = 1
    delta = m - n
    if abs(delta) > u:
        return None
    t = (u - abs(delta)) // 2
    d_lo, d_hi = min(0, delta) - t, max(0, delta) + t
    if trans:
        d_lo -= 1; d_hi += 1
    w = d_hi - d_lo + 1
    mask = (1 << w) - 1
    idx_ans = d_hi - delta
    # column 0, aligned to the window of column 1: row r = 1 - d_hi + i ; D[r,0] = |r|
    VP = VN = 0
    for i in range(w):
        r = 1 - d_hi + i
        if r >= 1: VP |= 1 << i
        else: VN |= 1 << i
    score = abs(delta)
    PMp = 0
    D0p = mask
    for j in range(1, m + 1):
        PM = 0
        for i in range(w):
            r = j - d_hi + i
            if 1 <= r <= n and a[r - 1] == b[j - 1]:
                PM |= 1 << i
        D0 = ((((PM & VP) + VP) ^ VP) | PM | VN) & mask
        if trans:
            D0 |= (~D0p & mask) & ((PM << 1) & mask) & (PMp >> 1)
        HP = (VN | ~(D0 | VP)) & mask
        HN = D0 & VP
        D0s = D0 >> 1
        VP = (HN | ~(D0s | HP)) & mask
        VN = D0s & HP
        score += 1 - ((D0 >> idx_ans) & 1)
        PMp, D0p = PM, D0
    return score if score <= k else None


if __name__ == "__main__":
    import datagen as Dg
    import oracle_lib as O
    g = Dg.rng(5)
    bad = 0
    tot = 0
    for it in range(3000):
        la = int(g.integers(0, 60))
        x = Dg.rand_str(g, la)
        typ = it % 4
        if typ == 0: y = Dg.rand_str(g, int(g.integers(0, 60)))
        elif typ == 1: y = x
        else: y = Dg.mutate(g, x, int(g.integers(0, 12)), True)
        for trans in (False, True):
            costs = (1, 1, 0, 1) if trans else (1, 1, 0, None)
            for k in (0, 1, 2, 3, 5, 8, 13, 40, 0xFFFFFFFF):
                want = O.levenshtein_simd_k_with_opts(x, y, k, False, costs)[0]
                got = bitpar_band(x, y, k, trans)
                tot += 1
                if got != want:
                    bad += 1
                    if bad < 10: print("MISMATCH", x, y, k, trans, got, want)
    print("checked", tot, "bad", bad)
