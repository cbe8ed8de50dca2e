"""Search tile function (lev_search_body.h) on the host: tiled + halo == monolithic scalar oracle (All mode),
for every cost family incl. transposition and affine gaps -- the empirical halo-equivalence proof SURVEY.md 8e asks for."""
import numpy as np
import pytest

import datagen as Dg
import emu_lib as E
import oracle_lib as O

COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (3, 1, 0, None), (1, 1, 2, None), (2, 1, 2, None), (2, 2, 1, 3), (1, 2, 0, 1)]


def oracle_all(needle, hay, k, costs, anchored=False):
    hits = O.levenshtein_search_naive_with_opts(needle, hay, k, O.ALL, costs, anchored)
    return [h for h in hits if h[1] > 0]      # the end == 0 match is a host special case


@pytest.mark.parametrize("costs", COSTS)
def test_tiled_equals_monolithic(costs):
    assert O.costs_valid(costs) and O.costs_valid_search(costs)
    g = Dg.rng(41)
    for n in (1, 3, 8, 13, 32):
        needle = Dg.rand_str(g, n)
        for k in (0, 1, n // 4 + 1, (n + 1) // 2):
            hay = Dg.planted_haystack(int(g.integers(1 << 30)), needle, 1500, 60, max(1, k))
            want = oracle_all(needle, hay, k, costs)
            for tile in (1 << 30, 64, 97, 256):
                tile = max(tile, 1)
                for packed in (False, True):
                    got = E.lev_search_tiled(needle, hay, k, costs, tile=tile, packed=packed)
                    assert got == want, (n, k, tile, costs, packed)


def test_small_alphabet_ties():
    """Binary-alphabet haystacks maximise cost ties, which is where the length tie rules (quirk Q2) bite."""
    g = Dg.rng(8)
    for costs in COSTS:
        for _ in range(30):
            n = int(g.integers(1, 9))
            needle = g.integers(97, 99, size=n, dtype=np.uint8).tobytes()
            hay = g.integers(97, 99, size=int(g.integers(0, 80)), dtype=np.uint8).tobytes()
            k = int(g.integers(0, n + 1))
            want = oracle_all(needle, hay, k, costs)
            for tile in (1 << 30, 16, 33):
                for packed in (False, True):
                    assert E.lev_search_tiled(needle, hay, k, costs, tile=tile, packed=packed) == want, (needle, hay, k, costs, tile, packed)


def test_anchored():
    g = Dg.rng(3)
    for costs in COSTS:
        for _ in range(30):
            n = int(g.integers(1, 12))
            needle = Dg.rand_str(g, n)
            hay = Dg.mutate(g, needle, 3, costs[3] is not None) + Dg.rand_str(g, 20)
            k = int(g.integers(0, 6))
            want = oracle_all(needle, hay, k, costs, anchored=True)
            for packed in (False, True):
                assert E.lev_search_tiled(needle, hay, k, costs, anchored=True, packed=packed) == want


def test_long_needles_memory_backed_column():
    """Needles beyond the register kernel's 32 rows use the memory-backed column (lev_search_tile_mem)."""
    g = Dg.rng(77)
    for costs in [(1, 1, 0, None), (1, 1, 0, 1), (2, 1, 2, None), (2, 2, 1, 3)]:
        for n in (33, 40, 97, 260):
            needle = Dg.rand_str(g, n)
            k = n // 5
            hay = Dg.planted_haystack(int(g.integers(1 << 30)), needle, 4 * n + 900, 3 * n, max(1, k))
            want = oracle_all(needle, hay, k, costs)
            for tile in (1 << 30, 257):
                assert E.lev_search_tiled(needle, hay, k, costs, tile=tile) == want, (n, k, tile, costs)


def test_anchored_large_k_packed_gate():
    """Anchored searches with a big k: row 0 costs (i+1)*gc + sg and may pass the packed form's "no gap yet" marker
    (0x7000).  Wherever the host's gate (srch_anchored_packed_ok) admits the packed form it must equal the oracle; the
    32+32-bit form must equal it everywhere.  (Round-1 advisor finding: needle "a", 30001 x 'b', k = 30000 under-reported.)"""
    cases = [(b"a", b"b" * 30001, 30000, (1, 1, 0, None)), (b"a", b"b" * 28000, 27000, (1, 1, 0, None)),
             (b"ab", b"b" * 200, 25000, (255, 255, 0, None)), (b"abc", b"c" * 150, 30000, (200, 255, 3, None)),
             (b"ab", b"b" * 120, 20000, (1, 200, 100, 2)), (b"abcd", b"xbcd" + b"q" * 100, 22000, (120, 250, 5, None))]
    admitted = 0
    for needle, hay, k, costs in cases:
        mc, gc, sg, tc = costs
        want = oracle_all(needle, hay, k, costs, anchored=True)
        assert E.lev_search_tiled(needle, hay, k, costs, anchored=True, packed=False) == want, (needle, k, costs)
        for cut in (len(hay), len(hay) // 2, 60, 20):
            sub = hay[:cut]
            h = min(len(sub), len(needle) + max(0, k - sg) // gc)
            if E.lib().emu_search_anchored_packed_ok(h, len(needle), mc, gc, sg):
                admitted += 1
                assert E.lev_search_tiled(needle, sub, k, costs, anchored=True, packed=True) == \
                    oracle_all(needle, sub, k, costs, anchored=True), (needle, cut, k, costs)
    assert admitted >= 6
    # the boundary itself: the longest haystack the gate admits for unit costs, and one column more (refused)
    needle, costs = b"a", (1, 1, 0, None)
    hmax = max(h for h in range(28000, 29000) if E.lib().emu_search_anchored_packed_ok(h, 1, 1, 1, 0))
    assert not E.lib().emu_search_anchored_packed_ok(hmax + 1, 1, 1, 1, 0)
    hay = b"b" * hmax
    assert E.lev_search_tiled(needle, hay, 30000, costs, anchored=True, packed=True) == oracle_all(needle, hay, 30000, costs, anchored=True)


@pytest.mark.parametrize("costs", COSTS)
def test_wave_per_block_equals_monolithic(costs):
    """lev_search_wave_body.h (one wavefront per block, lanes = needle rows, skewed columns, DPP hand-over) on the 64-lane
    host emulation: blocks + halo == the monolithic scalar oracle, needles up to 64 bytes, every cost family."""
    g = Dg.rng(77)
    mc, gc, sg, tc = costs
    for n in (1, 2, 3, 8, 13, 32, 33, 47, 64):
        needle = Dg.rand_str(g, n)
        for k in (0, 1, n // 4 + 1, (n + 1) // 2, max(0, n - 1)):
            halo = n + max(0, k - sg) // gc + 2
            hay = Dg.planted_haystack(int(g.integers(1 << 30)), needle, 900, 60, max(1, k))
            want = oracle_all(needle, hay, k, costs)
            for tile in (64, 16, 37):
                if tile + halo <= 256:
                    assert E.lev_search_tiled(needle, hay, k, costs, tile=tile, packed=2) == want, (n, k, tile, costs)


def test_wave_per_block_small_alphabet_ties():
    g = Dg.rng(9)
    for costs in COSTS:
        for _ in range(40):
            n = int(g.integers(1, 12))
            needle = g.integers(97, 99, size=n, dtype=np.uint8).tobytes()
            hay = g.integers(97, 99, size=int(g.integers(0, 120)), dtype=np.uint8).tobytes()
            k = int(g.integers(0, n + 1))
            want = oracle_all(needle, hay, k, costs)
            for tile in (64, 16, 33):
                assert E.lev_search_tiled(needle, hay, k, costs, tile=tile, packed=2) == want, (needle, hay, k, costs, tile)
