"""not-gpu: the bit-parallel candidate filter (lev_filter_body.h) flags EXACTLY the 64-column blocks that hold an
end position of cost <= k in the oracle's All-mode output -- so running the exact kernel on the flagged blocks only
loses nothing."""
import numpy as np
import pytest

import datagen as Dg
import emu_lib as E
import oracle_lib as O

LEV, RDAM = (1, 1, 0, None), (1, 1, 0, 1)


def oracle_blocks(needle, hay, k, costs):
    hits = O.levenshtein_search_naive_with_opts(needle, hay, k, O.ALL, costs, False)
    return sorted({(end - 1) // 64 for (_, end, _) in hits if end > 0})


@pytest.mark.parametrize("trans", [False, True])
def test_filter_blocks_equal_oracle_blocks(trans):
    g = Dg.rng(41)
    costs = RDAM if trans else LEV
    for n in (1, 2, 5, 16, 31, 32):
        needle = Dg.rand_str(g, n)
        hay = Dg.planted_haystack(100 + n, needle, 6000, 300 + 7 * n, max(1, n // 3))
        for k in sorted({0, 1, n // 4, n // 2, max(0, n - 1)}):
            for tile in (64, 256, 1024):
                got = E.lev_filter_blocks(needle, hay, k, trans, tile=tile)
                assert got == oracle_blocks(needle, hay, k, costs), (n, k, tile, trans)


def test_filter_small_alphabet_and_nulls():
    g = Dg.rng(42)
    for trans in (False, True):
        costs = RDAM if trans else LEV
        for n in (3, 8, 20, 32):
            needle = bytes(g.integers(0, 3, size=n).astype(np.uint8))
            hay = bytes(g.integers(0, 3, size=3000).astype(np.uint8))
            for k in (0, 1, n // 3):
                assert E.lev_filter_blocks(needle, hay, k, trans, tile=128) == oracle_blocks(needle, hay, k, costs), (n, k, trans)


@pytest.mark.parametrize("trans", [False, True])
def test_filter_long_needles(trans):
    """Needles of 33..512 bytes: the multi-dword form of the scan (and a short needle forced onto 2 and 3 dwords; beyond 256 bytes the
    vectors are 12 or 16 dwords whatever the needle's own count: round 5)."""
    g = Dg.rng(43)
    costs = RDAM if trans else LEV
    for n in (33, 64, 65, 100, 200, 256, 257, 300, 384, 385, 500, 512):
        needle = Dg.rand_str(g, n)
        hay = Dg.planted_haystack(200 + n, needle, 8000, 900 + n, max(1, n // 5))
        for k in (0, n // 6, n // 3):
            assert E.lev_filter_blocks(needle, hay, k, trans, tile=512) == oracle_blocks(needle, hay, k, costs), (n, k, trans)
    needle = bytes(g.integers(97, 100, size=40).astype(np.uint8))
    hay = bytes(g.integers(97, 100, size=3000).astype(np.uint8))
    for k in (5, 12, 20):
        assert E.lev_filter_blocks(needle, hay, k, trans, tile=128) == oracle_blocks(needle, hay, k, costs), (k, trans)
    needle = Dg.rand_str(g, 20)
    hay = Dg.planted_haystack(77, needle, 5000, 400, 6)
    for words in (2, 3):
        assert E.lev_filter_blocks(needle, hay, 7, trans, tile=256, words=words) == oracle_blocks(needle, hay, 7, costs)


@pytest.mark.parametrize("trans", [False, True])
def test_filter_lower_bound_form_is_a_superset(trans):
    """The product's scan settles the score once per 32 columns (lev_filter_step_h / lev_filter_fold32) and flags a block when
    a LOWER BOUND of its smallest cost is <= k: every oracle block must be flagged; on large-alphabet text the extra blocks
    are rare (here: only around the planted near-copies), on a binary alphabet they may be many -- never fewer."""
    g = Dg.rng(44)
    costs = RDAM if trans else LEV
    for n in (1, 2, 5, 16, 31, 32):
        needle = Dg.rand_str(g, n)
        hay = Dg.planted_haystack(300 + n, needle, 9000, 300 + 7 * n, max(1, n // 3))
        for k in sorted({0, 1, n // 4, n // 2, max(0, n - 1)}):
            want = oracle_blocks(needle, hay, k, costs)
            for tile in (64, 256, 1024):
                got = E.lev_filter_blocks(needle, hay, k, trans, tile=tile, words=-1)
                assert set(want) <= set(got), (n, k, tile, trans)
                if n >= 16 and k <= n // 2:          # extras only where the score dips: around the planted (mutated) copies
                    planted = len(hay) // (300 + 7 * n) + 1
                    assert len(got) <= len(want) + 2 * planted, (n, k, tile, len(got), len(want))
    for n in (3, 8, 20, 32):
        needle = bytes(g.integers(0, 2, size=n).astype(np.uint8))
        hay = bytes(g.integers(0, 2, size=4000).astype(np.uint8))
        for k in (0, 1, n // 3):
            assert set(oracle_blocks(needle, hay, k, costs)) <= set(E.lev_filter_blocks(needle, hay, k, trans, tile=128, words=-1))


WEIGHTED = [(2, 2, 0, None), (2, 3, 1, None), (2, 2, 1, 3), (3, 1, 0, None), (1, 2, 0, None), (1, 1, 2, None), (4, 3, 3, 5), (2, 2, 0, 2),
            (5, 5, 0, None), (3, 2, 0, 1), (1, 3, 1, 1)]


def test_filter_k_of_general_costs():
    """k' = max(k / min(mc, tc), (k - sg) / min(mc, gc, tc)); unit costs times g give k / g; unit costs give k."""
    assert E.search_filter_k(16, (1, 1, 0, None)) == 16 and E.search_filter_k(16, (1, 1, 0, 1)) == 16
    assert E.search_filter_k(16, (2, 2, 0, None)) == 8 and E.search_filter_k(17, (2, 2, 0, 2)) == 8
    assert E.search_filter_k(16, (2, 3, 1, None)) == 8          # no gap: 16 / 2; with a gap: 15 / 2 = 7
    assert E.search_filter_k(16, (2, 2, 1, 3)) == 8
    assert E.search_filter_k(16, (3, 1, 0, None)) == 16         # gaps of cost 1: sixteen of them fit
    assert E.search_filter_k(16, (3, 1, 6, None)) == 10         # ... but each run pays 6 first: (16 - 6) / 1 = 10 > 16 / 3
    assert E.search_filter_k(2, (3, 1, 6, None)) == 0
    assert E.search_filter_k(16, (4, 4, 0, 2)) == 8             # a cheap transposition counts as ONE unit edit


@pytest.mark.parametrize("costs", WEIGHTED)
def test_unit_filter_with_k_prime_is_a_superset_under_general_costs(costs):
    """Every 64-column block that holds a hit of the oracle's WEIGHTED search is flagged by the unit-cost scan run with
    k' = srch_filter_k (both scan forms, the halo the dispatch gives it: needle_len + k' + 2) -- planted near-copies,
    small alphabets (ties, many hits), swaps for the transposition families."""
    g = Dg.rng(4500 + sum(costs[:3]))
    trans = costs[3] is not None
    for n in (4, 9, 16, 32):
        needle = Dg.rand_str(g, n)
        hay = bytearray(Dg.planted_haystack(500 + n, needle, 7000, 280 + 5 * n, max(1, n // 4)))
        for pos in range(150, len(hay) - 2 * n, 560):                 # more planted copies, with adjacent swaps and gap runs
            m = bytearray(Dg.mutate(g, needle, max(1, n // 6), swaps=True))
            hay[pos:pos + len(m)] = m
        hay = bytes(hay)
        for k in sorted({1, costs[0], n // 2, n, 2 * n}):
            kf = E.search_filter_k(k, costs)
            if kf >= n:
                continue                                               # the dispatch runs no filter there
            want = oracle_blocks(needle, hay, k, costs)
            for words in (0, -1):
                got = E.lev_filter_blocks(needle, hay, kf, trans, tile=256, halo=n + kf + 2, words=words)
                assert set(want) <= set(got), (n, k, kf, costs, words, sorted(set(want) - set(got))[:5])
    # a small alphabet: hits everywhere, ties in every block
    needle = bytes(g.integers(97, 100, size=12).astype(np.uint8))
    hay = bytes(g.integers(97, 100, size=4000).astype(np.uint8))
    for k in (2, 4, 7, 11):
        kf = E.search_filter_k(k, costs)
        if kf < 12:
            got = E.lev_filter_blocks(needle, hay, kf, trans, tile=128, halo=12 + kf + 2, words=-1)
            assert set(oracle_blocks(needle, hay, k, costs)) <= set(got), (k, kf, costs)
