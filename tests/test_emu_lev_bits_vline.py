"""Kernel-logic check without a GPU: the VLINE form of the bit-parallel band kernel's fetch (lev_bits_body.h: whole 128-byte lines per
lane, per-lane band geometry and alignment, burst classes) as a 64-lane host emulation against the oracle, bit for bit.  LDS starts
out as 0xA5 garbage and every byte of a line outside the blobs reads as 0xA5: nothing may depend on either."""
import numpy as np
import pytest

import datagen as Dg
import emu_lib as E
import oracle_lib as O
from test_emu_lev_band import make_pairs, _edge_pairs

LEV, RDAM = (1, 1, 0, None), (1, 1, 0, 1)


def oracle(a, b, k, trans):
    costs = RDAM if trans else LEV
    return [O.levenshtein_simd_k_with_opts(x, y, k, False, costs)[0] for x, y in zip(a, b)]


@pytest.fixture(autouse=True)
def _vline():
    E.bits_vline(True)
    yield
    E.bits_vline(False)


@pytest.mark.parametrize("trans", [False, True])
def test_vline_small(trans):
    a, b = make_pairs(11, 200, 40, 6, trans)
    for k in (0, 1, 2, 3, 7, 12, 30, 32, 61, 64, 100, 0xFFFFFFFF):
        got, plan = E.lev_bits(a, b, k, trans)
        assert got == oracle(a, b, k, trans), (k, trans, plan)


@pytest.mark.parametrize("force_NA", [1, 2, 3, 5, 8, 9, 11, 16, 18, 24, 26, 32])
def test_vline_every_window_width(force_NA):
    for trans in (False, True):
        k = max(0, min(4 * force_NA - 1 - (2 if trans else 0), 9))
        a, b = make_pairs(100 + force_NA, 130, 170, max(1, k), trans)
        got, plan = E.lev_bits(a, b, k, trans, force_NA=force_NA)
        assert plan["NA"] == force_NA
        assert got == oracle(a, b, k, trans), (k, trans, plan)


@pytest.mark.parametrize("trans", [False, True])
def test_vline_band_edges(trans):
    """alignments along the edges of the narrow band; bands up to the widest window (deep warm-ups: the first line may lie wholly
    in front of the string)"""
    for u in (6, 13, 32, 50, 90, 120):
        a, b = _edge_pairs(0xB175 + u, 70, 300, u)
        for k in (u - 1, u, u + 1, min(125, 2 * u)):
            got, plan = E.lev_bits(a, b, k, trans)
            assert got == oracle(a, b, k, trans), (u, k, trans, plan)


@pytest.mark.parametrize("static", [1, 2, 3])
def test_vline_ragged_cfg2_shape(static):
    """lengths uniform on 32..256, b within +-4 of a, k = 32 (the ragged bench batch), in batch order and sorted by b's length
    (what the launcher's counting sort produces: every wavefront one column count); all three window forms"""
    g = Dg.rng(0x7A22 + static)
    n = 256
    la = g.integers(32, 257, size=n)
    lb = np.clip(la + g.integers(-4, 5, size=n), 1, 256)
    a = [Dg.random_bytes(g, int(x)).tobytes() for x in la]
    b = []
    for i in range(n):            # half of the pairs near (within k), half random
        if i % 2:
            s = bytearray(a[i][:int(lb[i])].ljust(int(lb[i]), b"q"))
            for p in g.integers(0, len(s), size=10):
                s[int(p)] = int(g.integers(1, 255))
            b.append(bytes(s))
        else:
            b.append(Dg.random_bytes(g, int(lb[i])).tobytes())
    want = oracle(a, b, 32, False)
    assert sum(x is not None for x in want) >= 100
    got, plan = E.lev_bits(a, b, 32, False, static=static)
    assert got == want, plan
    order = np.argsort(-lb, kind="stable")
    a2, b2 = [a[i] for i in order], [b[i] for i in order]
    got, plan = E.lev_bits(a2, b2, 32, False, static=static)
    assert got == [want[i] for i in order], plan


def test_vline_long_strings():
    """strings of many lines (700 bytes: six bursts per string), wide and narrow bands, both families"""
    a, b = make_pairs(0xC0, 64, 700, 40, True)
    for trans, k in [(False, 50), (True, 33), (False, 5), (True, 30), (False, 127)]:
        got, plan = E.lev_bits(a, b, k, trans)
        assert got == oracle(a, b, k, trans), (trans, k, plan)


def test_vline_null_bytes_and_degenerate():
    g = Dg.rng(77)
    a = [b"", b"", b"\0", b"\0\0\0", b"abc", b"\0a\0b", bytes(40), bytes(40), b"x", b"ab", b"ba", b"", bytes(130)]
    b = [b"", b"\0\0", b"\0", b"\0", b"", b"a\0b\0", bytes(37), bytes([0] * 20 + [1] + [0] * 19), b"y", b"ba", b"ab", bytes(129), b""]
    for _ in range(60):
        n = int(g.integers(0, 6))
        a.append(Dg.random_bytes(g, n).tobytes())
        b.append(Dg.random_bytes(g, int(g.integers(0, 6))).tobytes())
    for trans in (False, True):
        for k in (0, 1, 3, 32, 40, 100):
            got, plan = E.lev_bits(a, b, k, trans)
            assert got == oracle(a, b, k, trans), (k, trans, plan)


def test_vline_every_alignment_and_line_crossing():
    """one 200-byte pair at every start alignment 0..127 of both strings (the blob is shifted a byte at a time): the burst classes, the
    first-line rule and the ring addresses for each (s_a, s_b) residue"""
    g = Dg.rng(5)
    base_a = Dg.random_bytes(g, 200).tobytes()
    sb = bytearray(base_a)
    for p in (3, 77, 130, 199):
        sb[p] ^= 0x55
    del sb[50]
    base_b = bytes(sb)
    want = oracle([base_a], [base_b], 32, False)[0]
    assert want is not None
    for sh in range(0, 128, 1):
        pad_a, pad_b = Dg.random_bytes(g, sh).tobytes(), Dg.random_bytes(g, (sh * 7 + 5) % 128).tobytes()
        a = [pad_a, base_a] * 3
        b = [pad_b, base_b] * 3
        got, plan = E.lev_bits(a, b, 32, False)
        assert got[1] == want and got[3] == want and got[5] == want, (sh, got)
