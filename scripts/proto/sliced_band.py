"""Prototype of the pair-sliced band kernel's arithmetic: 32 pairs of equal lengths ride the 32 bits of a word; every band
cell is the unit-cost difference cell (Myers 1999, cell form): zero = Eq | Mv_in | Mh_in ...; the window slides one row per
column as in lev_bits_body.h; the answer is |delta| + columns - sum(zero on the answer diagonal).  Checked against the oracle."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo/tests")
import oracle_lib as O

M32 = 0xFFFFFFFF


def planes(strs, pos):
    """8 plane words for byte position pos of 32 strings (garbage-safe for out-of-range positions)."""
    out = [0] * 8
    for p, s in enumerate(strs):
        c = s[pos] if 0 <= pos < len(s) else (p * 37 + pos * 11) & 0xFF     # virtual rows: anything
        for k in range(8):
            out[k] |= ((c >> k) & 1) << p
    return out


def sliced(A, B, k, R=2):
    alen, blen = len(A[0]), len(B[0])
    delta = blen - alen
    uk = min(k, max(alen, blen))
    if abs(delta) > uk:
        return [None] * 32
    t = (uk - abs(delta)) // 2
    dhi, dlo = max(0, delta) + t, min(0, delta) - t
    W = dhi - dlo + 1
    S = -(-W // R)
    W = S * R                                    # widen downwards to whole strips
    w_ans = dhi - delta
    # column 0: rows r = w - dhi; dv(w) = +1 for r >= 1 else -1
    Pv = [M32 if (w - dhi) >= 1 else 0 for w in range(W)]
    Mv = [0 if (w - dhi) >= 1 else M32 for w in range(W)]
    cnt = [0] * 32
    for j in range(1, blen + 1):
        bp = planes(B, j - 1)
        nPv, nMv = [0] * W, [0] * W
        Ph, Mh = M32, 0                          # above the window: +1
        for w in range(W):
            r = j - dhi + w
            ap = planes(A, r - 1)
            neq = 0
            for q in range(8):
                neq |= ap[q] ^ bp[q]
            pv_in, mv_in = (Pv[w + 1], Mv[w + 1]) if w + 1 < W else (M32, 0)
            zero = (~neq | mv_in | Mh) & M32
            pv_out = (Mh | ~(zero | Ph)) & M32
            mv_out = zero & Ph
            ph_out = (mv_in | ~(zero | pv_in)) & M32
            mh_out = zero & pv_in
            nPv[w], nMv[w] = pv_out, mv_out
            Ph, Mh = ph_out, mh_out
            if w == w_ans:
                for p in range(32):
                    cnt[p] += (zero >> p) & 1
        Pv, Mv = nPv, nMv
    d = [abs(delta) + blen - c for c in cnt]
    return [x if x <= k else None for x in d]


def main():
    rng = np.random.default_rng(3)
    bad = 0
    for trial in range(60):
        alen = int(rng.integers(1, 70)); blen = max(1, alen + int(rng.integers(-6, 7)))
        k = int(rng.integers(1, 40))
        alpha = int(rng.choice([2, 4, 200]))
        A, B = [], []
        for p in range(32):
            a = rng.integers(1, 1 + alpha, alen, dtype=np.uint8)
            if rng.random() < 0.7:
                b = list(a)
                for _ in range(int(rng.integers(0, k + 3))):
                    op = rng.integers(0, 3)
                    if op == 0 and b: b[int(rng.integers(0, len(b)))] = int(rng.integers(1, 1 + alpha))
                    elif op == 1: b.insert(int(rng.integers(0, len(b) + 1)), int(rng.integers(1, 1 + alpha)))
                    elif b: del b[int(rng.integers(0, len(b)))]
                b = (b + list(rng.integers(1, 1 + alpha, blen, dtype=np.uint8)))[:blen]
            else:
                b = list(rng.integers(1, 1 + alpha, blen, dtype=np.uint8))
            A.append(bytes(a.tolist())); B.append(bytes(b))
        got = sliced(A, B, k, R=int(rng.choice([1, 2, 3, 4])))
        for p in range(32):
            want = O.levenshtein_naive_k_with_opts(A[p], B[p], k, False, O.LEVENSHTEIN_COSTS)
            want = None if want is None else want[0]
            if got[p] != want:
                bad += 1
                if bad < 6: print("MISMATCH", alen, blen, k, p, got[p], want)
    print("bad", bad)


if __name__ == "__main__":
    main()
