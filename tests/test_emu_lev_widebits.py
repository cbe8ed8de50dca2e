"""Kernel-logic check without a GPU: the row-blocked bit-parallel kernel body (lev_widebits_body.h: one pair per
wavefront, 32/64 rows per lane, nibble tables in LDS, skewed lanes) as a 64-lane host emulation against the oracle."""
import numpy as np
import pytest

import datagen as Dg
import emu_lib as E
import oracle_lib as O

LEV, RDAM = (1, 1, 0, None), (1, 1, 0, 1)


def oracle(a, b, k, trans):
    costs = RDAM if trans else LEV
    return [O.levenshtein_simd_k_with_opts(x, y, k, False, costs)[0] for x, y in zip(a, b)]


def pairs(seed, count, maxlen, small_alphabet=False):
    g = Dg.rng(seed)
    a, b = [], []
    for i in range(count):
        la = int(g.integers(0, maxlen + 1))
        if small_alphabet:
            x = bytes(g.integers(97, 100, size=la).astype(np.uint8))
        else:
            x = Dg.rand_str(g, la)
        t = i % 4
        if t == 0:
            lb = int(g.integers(0, maxlen + 1))
            y = bytes(g.integers(97, 100, size=lb).astype(np.uint8)) if small_alphabet else Dg.rand_str(g, lb)
        elif t == 1:
            y = x
        else:
            y = Dg.mutate(g, x, int(g.integers(0, 30)), True)
        a.append(x); b.append(y)
    return a, b


@pytest.mark.parametrize("nwl", [1, 2])
@pytest.mark.parametrize("trans", [False, True])
def test_widebits_ragged(nwl, trans):
    a, b = pairs(5 + nwl, 60, 300)
    for k in (0, 3, 25, 100, 0xFFFFFFFF):
        got = E.lev_widebits(a, b, k, trans, nwl=nwl)
        assert got == oracle(a, b, k, trans), (k, trans, nwl)


@pytest.mark.parametrize("nwl", [1, 2])
def test_widebits_small_alphabet_transpositions(nwl):
    """Three-letter alphabet: matches and transpositions everywhere, incl. across lane (row-block) boundaries."""
    a, b = pairs(50 + nwl, 50, 200, small_alphabet=True)
    for trans in (False, True):
        got = E.lev_widebits(a, b, 0xFFFFFFFF, trans, nwl=nwl)
        assert got == oracle(a, b, 0xFFFFFFFF, trans), (trans, nwl)


def test_widebits_many_lanes_and_chunks():
    """Rows spread over all 64 lanes (2000+ bytes) and columns over many 64-byte chunks; null bytes; swapped roles."""
    g = Dg.rng(9)
    x = Dg.rand_str(g, 2048)
    y = Dg.mutate(g, x, 150, True)
    z = bytes(2000)
    w = bytes([0] * 700 + [7] + [0] * 1299)
    a = [x, y, x, z, w, x[:33], b"", x]
    b = [y, x, Dg.rand_str(g, 1900), w, z, x[:2000], x[:100], x]
    for trans in (False, True):
        got = E.lev_widebits(a, b, 0xFFFFFFFF, trans, nwl=1, nwaves=2)
        assert got == oracle(a, b, 0xFFFFFFFF, trans), trans
    got = E.lev_widebits(a, b, 140, False, nwl=2, nwaves=1)
    assert got == oracle(a, b, 140, False)


def test_widebits_4k_pair():
    """BASELINE cfg3 geometry: one 4 KiB pair, 64 rows per lane."""
    g = Dg.rng(0x7A03)
    x, y = Dg.rand_str(g, 4096), Dg.rand_str(g, 4096)
    ym = Dg.mutate(g, x, 500, False)
    got = E.lev_widebits([x, x], [y, ym], 0xFFFFFFFF, False, nwl=2, nwaves=1)
    assert got == [O.levenshtein(x, y), O.levenshtein(x, ym)]


def test_widebits_several_stripes_and_band_limited_columns():
    """Both strings longer than one stripe (2048 rows at 32 rows per lane): boundary lines between stripes, the anchor
    chain, and -- with a bounded k -- stripes that only visit the columns of their band."""
    g = Dg.rng(0x57)
    x = Dg.rand_str(g, 5000)
    y = Dg.mutate(g, x, 120, True)
    z = Dg.rand_str(g, 4500)
    w = bytes(g.integers(97, 100, size=4300).astype(np.uint8))
    v = bytes(g.integers(97, 100, size=4100).astype(np.uint8))
    a = [x, y, x, w, x[:2049], x[:2048], x]
    b = [y, x, z, v, y[:2100], y[:2048], x]
    for trans in (False, True):
        for k in (0xFFFFFFFF, 300, 130):
            got = E.lev_widebits(a, b, k, trans, nwl=1, nwaves=2)
            assert got == oracle(a, b, k, trans), (trans, k)
    got = E.lev_widebits(a[:3], b[:3], 0xFFFFFFFF, True, nwl=2, nwaves=1)      # 4096-row stripes: 2 stripes
    assert got == oracle(a[:3], b[:3], 0xFFFFFFFF, True)


@pytest.mark.parametrize("trans", [False, True])
def test_widebits_huge_pair_as_tiles(trans):
    """The many-wavefront form for ONE long pair: stripes x tiles launched diagonal by diagonal.  Every tile size and
    both tile orders inside a launch must reproduce the oracle -- a wrong dependency offset between the stripes'
    tile grids shows up as a read of an unwritten boundary entry (the buffers are poisoned)."""
    g = Dg.rng(0x4867)
    x = Dg.rand_str(g, 6500)
    y = Dg.mutate(g, x, 200, True)
    z = bytes(g.integers(97, 100, size=5000).astype(np.uint8))
    w = bytes(g.integers(97, 100, size=4600).astype(np.uint8))
    cases = [(x, y, 0xFFFFFFFF), (y, x, 400), (x, y, 230), (z, w, 0xFFFFFFFF), (x[:2048], y[:2048], 0xFFFFFFFF), (x[:2049], y[:3000], 0xFFFFFFFF),
             (x, x[:6000], 600), (x[:100], y[:4000], 0xFFFFFFFF)]
    for a, b, k in cases:
        want = oracle([a], [b], k, trans)[0]
        for tile_steps, order in [(64, 0), (256, 1), (1024, 0), (4096, 1)]:
            assert E.lev_widebits_huge(a, b, k, trans, nwl=1, tile_steps=tile_steps, order=order) == want, (len(a), len(b), k, tile_steps, order)
    assert E.lev_widebits_huge(x, y, 0xFFFFFFFF, trans, nwl=2, tile_steps=512, order=1) == oracle([x], [y], 0xFFFFFFFF, trans)[0]


@pytest.mark.parametrize("trans", [False, True])
def test_widebits_traceback_equals_the_scalar_traceback(trans):
    """TRACE form + host walk: the run-length edit script must equal the oracle's (the scalar path's tie order) -- random,
    mutated, small-alphabet (ties everywhere, transpositions across lane boundaries), swapped roles, several stripes,
    bounded k (band-limited stripes)."""
    costs = RDAM if trans else LEV
    g = Dg.rng(0x7ACE)
    x = Dg.rand_str(g, 700)
    y = Dg.mutate(g, x, 60, True)
    s1 = bytes(g.integers(97, 100, size=300).astype(np.uint8))
    s2 = bytes(g.integers(97, 100, size=280).astype(np.uint8))
    xl = Dg.rand_str(g, 4500)
    yl = Dg.mutate(g, xl, 150, True)
    cases = [(x, y, 0xFFFFFFFF, 1), (y, x, 0xFFFFFFFF, 1), (s1, s2, 0xFFFFFFFF, 1), (s2, s1, 0xFFFFFFFF, 2), (x, y, 150, 2), (b"", b"abc", 9, 1),
             (b"abc", b"", 9, 1), (b"ab", b"ba", 9, 1), (x[:33], y[:70], 0xFFFFFFFF, 1), (xl, yl, 0xFFFFFFFF, 1), (xl, yl, 400, 1)]
    for a, b, k, nwl in cases:
        want = O.levenshtein_simd_k_with_opts(a, b, k, True, costs)
        got = E.lev_widebits_trace(a, b, k, trans, nwl=nwl)
        assert got[0] == want[0], (len(a), len(b), k, nwl)
        assert got[1] == want[1], (len(a), len(b), k, nwl, got[1][:6], want[1][:6])
    assert E.lev_widebits_trace(x, y, 20, trans) == (None, None)
