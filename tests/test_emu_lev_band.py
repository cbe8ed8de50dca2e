"""Kernel-logic check without a GPU: the band-wavefront kernel body (lev_band_body.h) run as a 64-lane
host emulation must equal the oracle's scalar banded path bit for bit, for every lane layout (D, L),
cost family, ragged lengths and multi-chunk strings.  The same body is what hipcc compiles for gfx950."""
import numpy as np
import pytest

import datagen as Dg
import emu_lib as E
import oracle_lib as O

COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (2, 3, 0, None), (3, 1, 0, None), (1, 1, 2, None), (2, 1, 2, None),
         (2, 2, 1, 3), (5, 3, 4, 4)]


def oracle(a, b, k, costs):
    return [O.levenshtein_simd_k_with_opts(x, y, k, False, costs)[0] for x, y in zip(a, b)]


def make_pairs(seed, n, maxlen, kmut, swaps):
    g = Dg.rng(seed)
    a, b = [], []
    for i in range(n):
        la = int(g.integers(0, maxlen + 1))
        x = Dg.rand_str(g, la)
        t = i % 4
        if t == 0:
            y = Dg.rand_str(g, int(g.integers(0, maxlen + 1)))
        elif t == 1:
            y = x
        else:
            y = Dg.mutate(g, x, kmut, swaps)
        a.append(x); b.append(y)
    return a, b


@pytest.mark.parametrize("costs", COSTS)
def test_emu_small_all_costs(costs):
    a, b = make_pairs(3, 150, 40, 6, costs[3] is not None)
    for k in (0, 1, 3, 7, 12, 30):
        got, plan = E.lev_band(a, b, k, costs)
        assert got == oracle(a, b, k, costs), (k, costs, plan)


@pytest.mark.parametrize("force_D,force_L", [(2, 0), (4, 0), (6, 0), (8, 3), (10, 2), (12, 0), (16, 4), (18, 1),
                                             (22, 3), (24, 0), (34, 2), (66, 1), (8, 9), (2, 40)])
def test_emu_lane_layouts(force_D, force_L):
    """cfg2-like band (k chosen so the needed diagonals fit the forced layout) on every (D, L) shape."""
    cap = force_D * (force_L if force_L else 64)
    k = min(32, max(0, (cap - 2) // 2))
    for costs in [(1, 1, 0, None), (1, 1, 0, 1), (2, 2, 1, 3)]:
        a, b = make_pairs(7 + force_D, 70, 90, max(1, k), costs[3] is not None)
        got, plan = E.lev_band(a, b, k, costs, force_D=force_D, force_L=force_L)
        assert plan["D"] == force_D
        assert got == oracle(a, b, k, costs), (k, costs, plan)


def test_emu_cfg2_shape():
    """BASELINE cfg2 geometry: 256 B pairs, k = 32, LEVENSHTEIN_COSTS; random (all None) + mutated."""
    ar, br = Dg.pairs_random(0x7A02, 30, 256)
    am, bm = Dg.pairs_mutated_fixed(0x7A12, 50, 256, 32)
    a = [x.tobytes() for x in ar] + [x.tobytes() for x in am]
    b = [x.tobytes() for x in br] + [x.tobytes() for x in bm]
    got, plan = E.lev_band(a, b, 32)
    want = oracle(a, b, 32, (1, 1, 0, None))
    assert got == want
    assert any(v is not None for v in want) and any(v is None for v in want)


def test_emu_cfg4_shape():
    """BASELINE cfg4 geometry: 128 B pairs, k = 8, RDAMERAU_COSTS with planted adjacent swaps."""
    am, bm = Dg.pairs_mutated_fixed(0x7A04, 80, 128, 8, swaps=True)
    a = [x.tobytes() for x in am]; b = [x.tobytes() for x in bm]
    got, plan = E.lev_band(a, b, 8, O.RDAMERAU_COSTS)
    assert got == oracle(a, b, 8, O.RDAMERAU_COSTS)
    lev = oracle(a, b, 8, O.LEVENSHTEIN_COSTS)
    assert got != lev  # the transposition path must matter on this data


def test_emu_long_multichunk():
    """Strings spanning several 64-byte stream chunks, ragged, incl. length difference > band (early None)."""
    g = Dg.rng(99)
    a, b = [], []
    for n in (63, 64, 65, 127, 128, 129, 300, 500, 1000):
        x = Dg.rand_str(g, n)
        a += [x, x, x]
        b += [Dg.mutate(g, x, 20), x[: n // 2], Dg.rand_str(g, n + 5)]
    for k, costs in [(20, (1, 1, 0, None)), (25, (1, 1, 0, 1)), (40, (2, 1, 3, None)), (0xFFFFFFFF, (1, 1, 0, None))]:
        if k == 0xFFFFFFFF:
            aa, bb = a[:12], b[:12]   # full-matrix band: keep it small
        else:
            aa, bb = a, b
        got, plan = E.lev_band(aa, bb, k, costs)
        assert got == oracle(aa, bb, k, costs), (k, costs, plan)


def test_emu_null_bytes_and_edges():
    """Zero is the window filler value, so NUL bytes are the edge case the reference tests for its SIMD path
    (tests/basic_tests.rs:503-537)."""
    a = [b"\0", b"ab\0de", b"\0b", b"\0", b"\0", b"\0\0b\0", b"x", b"", b"a" * 70, b"\0" * 40]
    b = [b"", b"a\0bde", b"b\0", b"\0\0", b"\0", b"\0b\0\0", b"x\0", b"\0\0\0", b"", b"\0" * 37 + b"a"]
    for costs in [(1, 1, 0, None), (1, 1, 0, 1), (2, 2, 1, 3)]:
        for k in (0, 1, 2, 5, 100):
            got, plan = E.lev_band(a, b, k, costs)
            assert got == oracle(a, b, k, costs), (k, costs, plan)


def test_emu_affine_kernel_on_linear_costs():
    """The AFFINE instantiation must agree with the linear one when start_gap = 0."""
    a, b = make_pairs(5, 100, 60, 8, True)
    for costs in [(1, 1, 0, None), (1, 1, 0, 1), (3, 2, 0, None)]:
        g1, _ = E.lev_band(a, b, 9, costs)
        g2, _ = E.lev_band(a, b, 9, costs, force_affine=True)
        assert g1 == g2 == oracle(a, b, 9, costs)


def test_emu_transposition_forms_agree():
    """The dot4-penalty form and the select form of the transposition must agree; big mismatch costs
    (2*mc > 255 + tc) take the select form by themselves."""
    a, b = make_pairs(9, 120, 50, 8, True)
    for costs in [(1, 1, 0, 1), (2, 2, 1, 3), (100, 90, 3, 150), (200, 130, 0, 255), (255, 255, 255, 255)]:
        assert O.costs_valid(costs)
        for k in (3, 40, 700):
            want = oracle(a, b, k, costs)
            assert E.lev_band(a, b, k, costs)[0] == want, (costs, k)
            assert E.lev_band(a, b, k, costs, force_trans_select=True)[0] == want, (costs, k)


def _edge_pairs(seed, n, base_len, u):
    """Pairs whose optimal alignment hugs the edge of the narrow band: a block deleted near the start and
    re-inserted near the end (and the mirror image), on top of a length difference, so the path first strays
    t diagonals to one side and only returns at the very end."""
    g = Dg.rng(seed)
    a, b = [], []
    for i in range(n):
        x = bytearray(Dg.rand_str(g, base_len))
        delta = int(g.integers(0, u + 1))
        t = int(g.integers(0, (u - delta) // 2 + 2))           # sometimes one more than the band allows -> None
        y = bytearray(x)
        blk = bytes(Dg.rand_str(g, t))
        y = y[t:] if i & 1 else bytearray(blk) + y              # stray t diagonals right at the start ...
        y = y + bytearray(blk) if i & 1 else y[:len(y) - t]     # ... and come back at the very end
        y = y + bytearray(Dg.rand_str(g, delta))                # plus the length difference
        pair = (bytes(x), bytes(y))
        if i & 2:
            pair = pair[::-1]
        a.append(pair[0]); b.append(pair[1])
    return a, b


@pytest.mark.parametrize("costs", [(1, 1, 0, None), (9, 1, 0, None), (3, 2, 0, 2), (2, 1, 3, None), (255, 1, 0, None)])
def test_emu_narrow_band_edges(costs):
    """The kernel's band is [min(0,delta) - t, max(0,delta) + t], t = (unit_k - |delta|)/2 -- about half of the
    reference's [-unit_k, unit_k] (lev_plan.h).  Alignments built to run along its edges must still come out
    exactly as the scalar path reports them, for every k around the true distance."""
    for u in (6, 13, 32):
        a, b = _edge_pairs(0xED6E + u, 60, 70, u)
        k_unit = costs[1] * u + costs[2]
        for k in (k_unit - 1, k_unit, k_unit + costs[1], 2 * k_unit + 1, 0xFFFFFFFF):
            got, plan = E.lev_band(a, b, max(k, 0), costs)
            assert got == oracle(a, b, max(k, 0), costs), (u, k, costs, plan)


@pytest.mark.parametrize("chunk", [16, 32, 64])
def test_emu_stream_chunk_lengths(chunk):
    """The LDS ring works for every chunk length the planner may pick, on strings spanning many chunks."""
    a, b = make_pairs(0xC4 + chunk, 64, 300, 20, True)
    for costs, k in [((1, 1, 0, None), 25), ((1, 1, 0, 1), 40), ((2, 1, 2, None), 30)]:
        for D, L in [(0, 0), (4, 0), (24, 2)]:
            got, plan = E.lev_band(a, b, k, costs, force_D=D, force_L=L, chunk=chunk)
            assert got == oracle(a, b, k, costs), (chunk, costs, k, plan)


# ---- round 3: the score form (cells hold gc (i+j) - dp, maxima instead of minima) against the cost form and the oracle

SCORE_COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (2, 3, 1, None), (2, 2, 1, 3), (2, 1, 0, None), (1, 2, 0, None), (254, 127, 0, None),
               (254, 127, 255, 253), (3, 2, 0, 2), (1, 127, 7, None), (4, 2, 255, None), (2, 1, 3, 1)]


@pytest.mark.parametrize("costs", SCORE_COSTS)
def test_emu_score_form_against_cost_form(costs):
    assert O.costs_valid(costs) and E.lev_band_score_applies(costs)
    a, b = make_pairs(31 + costs[0], 130, 70, 7, costs[3] is not None)
    a += [b"", b"x", b"", b"abc" * 30, b"\0" * 50]
    b += [b"", b"", b"yy", b"abc" * 29 + b"ab", b"\0" * 47]
    for k in (0, 3, costs[1] * 6 + costs[2], 40, 300, 0xFFFFFFFF):
        want = oracle(a, b, k, costs)
        g_score, plan = E.lev_band(a, b, k, costs)
        g_cost, _ = E.lev_band(a, b, k, costs, score=False)
        assert g_score == want, (k, costs, plan)
        assert g_cost == want, (k, costs, plan)


@pytest.mark.parametrize("force_D,force_L", [(2, 8), (4, 4), (6, 1), (12, 1), (12, 2), (34, 1), (16, 64)])
def test_emu_score_form_lane_layouts(force_D, force_L):
    k = min(force_D * force_L - 2, 60)
    for costs in [(2, 3, 1, None), (2, 2, 1, 3), (1, 1, 0, None), (2, 1, 0, 2)]:
        ku = k * costs[1] + costs[2]
        a, b = make_pairs(11 + force_D, 80, 75, max(1, k // 2), costs[3] is not None)
        got, plan = E.lev_band(a, b, ku, costs, force_D=force_D, force_L=force_L)
        assert (plan["D"], plan["L"]) == (force_D, force_L)
        assert got == oracle(a, b, ku, costs), (ku, costs, plan)
        assert E.lev_band(a, b, ku, costs, force_D=force_D, force_L=force_L, score=False)[0] == got


def test_emu_score_form_rule():
    """2 gc - mc and 2 gc are bytes of a v_dot4: the rule that keeps everything else on the cost form"""
    assert not E.lev_band_score_applies((3, 1, 0, None))          # a mismatch dearer than two gap steps (possible with affine gaps)
    assert not E.lev_band_score_applies((1, 128, 0, None))        # 2 gc > 255
    assert not E.lev_band_score_applies((200, 130, 0, 255))
    assert not E.lev_band_score_applies((2, 2, 1, 3), force_trans_select=True)
    assert not E.lev_band_score_applies((120, 100, 0, 100))      # the transposition's gain 4 gc - tc = 300 is not a byte
    assert E.lev_band_score_applies((120, 100, 0, 145))
    assert E.lev_band_score_applies((254, 127, 0, None)) and E.lev_band_score_applies((2, 1, 9, 0 + 1))
    # and the costs the rule excludes still come out right (cost form)
    a, b = make_pairs(5, 90, 60, 5, False)
    for costs in [(3, 1, 0, None), (1, 128, 0, None), (255, 127, 3, None)]:
        assert E.lev_band(a, b, 900, costs)[0] == oracle(a, b, 900, costs)
    a, b = make_pairs(6, 90, 60, 5, True)
    for costs in [(120, 100, 0, 100), (120, 100, 0, 145), (120, 100, 7, 199)]:
        assert O.costs_valid(costs)
        assert E.lev_band(a, b, 2000, costs)[0] == oracle(a, b, 2000, costs)


@pytest.mark.parametrize("costs", [(2, 3, 1, None), (2, 2, 1, 3), (2, 3, 0, None), (1, 1, 0, None), (1, 1, 0, 1), (3, 2, 2, 2)])
def test_emu_band_line_form(costs):
    """The LINE form of the DP band kernel's fetch (one lane per pair, fixed-length batches: whole 128-byte lines parked in registers and
    committed to the ring piece by piece) against the oracle and against the chunk form: string lengths below / at / above one and two
    lines, different lengths of a and b (band offsets on both sides), k that keep the band in one lane (L = 1)."""
    if not (O.costs_valid(costs) and E.lev_band_score_applies(costs)):
        pytest.skip("costs outside the score form")
    g = Dg.rng(0xB1 + costs[0] * 7 + costs[1])
    for la, lb in [(256, 256), (128, 128), (100, 100), (260, 250), (250, 260), (129, 127), (16, 16), (15, 17), (300, 300), (48, 64), (1, 1)]:
        n = 70
        a, b = [], []
        for i in range(n):
            x = Dg.rand_str(g, la)
            y = (Dg.mutate(g, x, 6, costs[3] is not None)[:lb]).ljust(lb, b"q") if i % 3 else Dg.rand_str(g, lb)
            a.append(x); b.append(y)
        for k in (8, 20, 32):
            want = oracle(a, b, k, costs)
            E.band_line(True)
            try:
                got, plan = E.lev_band(a, b, k, costs)
            finally:
                E.band_line(False)
            chunk, plan2 = E.lev_band(a, b, k, costs)
            assert plan == plan2
            assert got == want, (la, lb, k, costs, plan)
            assert chunk == want
