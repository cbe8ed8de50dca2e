"""Cross-implementation property checks the reference runs before its benches
(benches/rand_benchmarks.rs:17-21, 45-46, 65-67, 88-90, 113-114), on the oracle's two
independent restatements (full matrix vs banded), extended to affine and transposition costs."""
import numpy as np
import pytest

import datagen as D
import oracle_lib as O

COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (2, 3, 0, None), (3, 1, 0, None), (1, 1, 2, None), (2, 1, 2, None),
         (2, 2, 1, 3), (5, 3, 4, 4), (1, 2, 0, 1)]


def py_distance(a, b, costs):
    """Independent pure-Python Gotoh/Damerau DP (textbook form, not following the reference text)."""
    mc, gc, sg, tc = costs
    INF = 10 ** 9
    n, m = len(a), len(b)
    H = [[INF] * (m + 1) for _ in range(n + 1)]
    E = [[INF] * (m + 1) for _ in range(n + 1)]
    F = [[INF] * (m + 1) for _ in range(n + 1)]
    H[0][0] = 0
    for i in range(n + 1):
        for j in range(m + 1):
            if i == 0 and j == 0:
                continue
            if j > 0:
                E[i][j] = min(H[i][j - 1] + sg + gc, E[i][j - 1] + gc)
            if i > 0:
                F[i][j] = min(H[i - 1][j] + sg + gc, F[i - 1][j] + gc)
            best = min(E[i][j], F[i][j])
            if i > 0 and j > 0:
                best = min(best, H[i - 1][j - 1] + (mc if a[i - 1] != b[j - 1] else 0))
                if tc is not None and i > 1 and j > 1 and a[i - 1] == b[j - 2] and a[i - 2] == b[j - 1]:
                    best = min(best, H[i - 2][j - 2] + tc)
            H[i][j] = best
    return H[n][m]


@pytest.mark.parametrize("costs", COSTS)
def test_full_vs_python(costs):
    assert O.costs_valid(costs)
    g = D.rng(11)
    for _ in range(150):
        la, lb = int(g.integers(0, 14)), int(g.integers(0, 14))
        a = g.integers(97, 100, size=la, dtype=np.uint8).tobytes()
        b = g.integers(97, 100, size=lb, dtype=np.uint8).tobytes()
        assert O.levenshtein_naive_with_opts(a, b, False, costs)[0] == py_distance(a, b, costs)


@pytest.mark.parametrize("costs", COSTS)
def test_banded_equals_full(costs):
    g = D.rng(1234)
    for it in range(200):
        la = int(g.integers(0, 60))
        a = D.rand_str(g, la)
        b = D.mutate(g, a, int(g.integers(0, 12)), swaps=costs[3] is not None) if it % 3 else D.rand_str(g, int(g.integers(0, 60)))
        full = O.levenshtein_naive_with_opts(a, b, False, costs)[0]
        assert O.levenshtein_simd_k_with_opts(a, b, 0xFFFFFFFF, False, costs)[0] == full
        for k in (0, 1, 2, 5, 9, 17, 40, 300):
            got = O.levenshtein_naive_k_with_opts(a, b, k, False, costs)[0]
            assert got == (full if full <= k else None), (a, b, k, costs)


def test_exp_equals_levenshtein():
    g = D.rng(5)
    for n in (10, 100, 300):
        for _ in range(20):
            a = D.rand_str(g, n)
            b = D.mutate(g, a, n // 10)
            d = O.levenshtein_naive_with_opts(a, b)[0]
            assert O.levenshtein(a, b) == d == O.levenshtein_exp(a, b)
            assert O.rdamerau(a, b) == O.rdamerau_exp(a, b) == O.levenshtein_naive_with_opts(a, b, False, O.RDAMERAU_COSTS)[0]


def test_trace_replays_to_cost():
    """A traceback must be a valid edit script of exactly the reported cost (both restatements)."""
    g = D.rng(77)
    for costs in COSTS:
        mc, gc, sg, tc = costs
        for _ in range(60):
            a = D.rand_str(g, int(g.integers(0, 25)))
            b = D.mutate(g, a, 6, swaps=tc is not None)
            for fn in (lambda: O.levenshtein_naive_with_opts(a, b, True, costs),
                       lambda: O.levenshtein_naive_k_with_opts(a, b, 0xFFFFFFFF, True, costs)):
                d, tr = fn()
                i = j = cost = 0
                for e, c in tr:
                    if e in ("Match", "Mismatch"):
                        for _ in range(c):
                            assert (a[i] == b[j]) == (e == "Match")
                            cost += 0 if e == "Match" else mc
                            i += 1; j += 1
                    elif e == "AGap":      # gap in a: consumes b
                        cost += sg + c * gc; j += c
                    elif e == "BGap":
                        cost += sg + c * gc; i += c
                    else:
                        for _ in range(c):
                            assert a[i] == b[j + 1] and a[i + 1] == b[j]
                            cost += tc; i += 2; j += 2
                assert (i, j) == (len(a), len(b))
                # one traceback code per cell carries no gap state, so with affine gaps (sg > 0) the
                # reference's script may re-open a gap it was extending: its cost can only be >= d.
                assert cost == d if sg == 0 else cost >= d


def test_hamming_and_search_properties():
    g = D.rng(9)
    for n in (1, 10, 100, 1000):
        a = D.rand_str(g, n)
        b = bytearray(a)
        for p in g.choice(n, size=n // 10, replace=False):
            b[p] = 32
        assert O.hamming_naive(a, bytes(b)) == sum(x != y for x, y in zip(a, b))
    assert O.hamming_naive(b"ab", b"abc") is None
    needle = D.rand_str(g, 10)
    hay = D.planted_haystack(3, needle, 2000, 100, 2)
    allm = O.hamming_search_naive_with_opts(needle, hay, 3, O.ALL)
    assert allm == O.hamming_search_simd_with_opts(needle, hay, 3, O.ALL)
    for s, e, k in allm:
        assert e - s == 10 and sum(x != y for x, y in zip(needle, hay[s:e])) == k <= 3
    best = O.hamming_search_simd_with_opts(needle, hay, 3, O.BEST)
    assert best and all(m[2] == min(x[2] for x in allm) for m in best)
    with pytest.raises(ValueError):
        O.hamming_search_simd_with_opts(b"ab", b"a\x00b", 1, O.ALL)


def test_search_matches_are_real_alignments():
    """Every All-mode hit (start,end,k) must satisfy distance(needle, haystack[start:end]) == k and be the
    minimum over all substrings ending at `end`."""
    g = D.rng(21)
    for costs in [(1, 1, 0, None), (1, 1, 0, 1), (1, 1, 2, None), (2, 1, 2, None)]:
        needle = D.rand_str(g, 8)
        hay = D.planted_haystack(4, needle, 400, 40, 3)
        k = 3
        hits = O.levenshtein_search_naive_with_opts(needle, hay, k, O.ALL, costs, False)
        ends = {e: (s, c) for s, e, c in hits}
        for e in range(1, len(hay) + 1):
            best = min(py_distance(needle, hay[s:e], costs) for s in range(max(0, e - 20), e + 1))
            if best <= k:
                assert e in ends and ends[e][1] == best, (e, best, costs)
                s = ends[e][0]
                assert py_distance(needle, hay[s:e], costs) == best
            else:
                assert e not in ends


def test_select_ladder_on_baseline_configs():
    # dispatch outcomes at the BASELINE configs
    assert O.levenshtein_select(256, 256, 32) == (32, 32, 8, 64)
    assert O.levenshtein_select(128, 128, 8, O.RDAMERAU_COSTS) == (8, 8, 8, 32)
    assert O.levenshtein_select(4096, 4096, 30)[2:] == (8, 32)
    assert O.levenshtein_select(4096, 4096, 240)[2:] == (8, 256)
    assert O.levenshtein_select(4096, 4096, 480)[2:] == (16, 0)
    assert O.levenshtein_select(4096, 4096, 7680) == (4096, 4096, 16, 0)
    assert O.levenshtein_select(70000, 70000, 0xFFFFFFFF)[2] == 32
    assert O.band_cells(256, 256, 32) == 15584
    assert O.band_cells(128, 128, 8, O.RDAMERAU_COSTS) == 2104
    assert O.band_cells(4096, 4096, 0xFFFFFFFF) == 4096 * 4096


def test_scalar_choice_where_the_simd_blend_differs():
    """The oracle (and so the product) returns the SCALAR path's value on the inputs where the reference's SIMD transposition
    blend gives another one; the blend model reproduces the listed SIMD-side values, so the split is deliberate and visible."""
    from scalar_vs_simd_cases import CASES, blend_model
    for a, b, costs, scalar, blend in CASES:
        assert O.levenshtein_naive_with_opts(a, b, False, costs)[0] == scalar
        assert O.levenshtein_naive_k_with_opts(a, b, 10, False, costs)[0] == scalar
        assert O.levenshtein_simd_k_with_opts(a, b, 10, False, costs)[0] == scalar        # the public-contract restatement: scalar rules
        assert blend_model(a, b, costs[0], costs[1], costs[3]) == blend != scalar
