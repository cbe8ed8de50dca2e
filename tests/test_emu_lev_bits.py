"""Kernel-logic check without a GPU: the bit-parallel band kernel body (lev_bits_body.h, unit-cost families) run as
a 64-lane host emulation must equal the oracle's scalar banded path bit for bit -- ragged lengths, every window
width, both cost families, strings spanning many LDS chunks, null bytes."""
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


@pytest.mark.parametrize("trans", [False, True])
def test_bits_small(trans):
    a, b = make_pairs(11, 200, 40, 6, trans)
    for k in (0, 1, 2, 3, 7, 12, 30, 61, 64, 100, 0xFFFFFFFF):
        got, plan = E.lev_bits(a, b, k, trans)
        assert got == oracle(a, b, k, trans), (k, trans, plan)


@pytest.mark.parametrize("force_NA", [1, 2, 3, 5, 8, 9, 11, 16, 18, 24, 26, 32])
def test_bits_every_window_width(force_NA):
    """The same small band inside every window width (1 or 2 dwords per bit-vector, partial top dwords)."""
    for trans in (False, True):
        k = max(0, min(4 * force_NA - 1 - (2 if trans else 0), 9))
        a, b = make_pairs(100 + force_NA, 130, 70, max(1, k), trans)
        got, plan = E.lev_bits(a, b, k, trans, force_NA=force_NA)
        assert plan["NA"] == force_NA
        assert got == oracle(a, b, k, trans), (k, trans, plan)


@pytest.mark.parametrize("trans", [False, True])
def test_bits_band_edges(trans):
    """Alignments that run along the edges of the narrow band, k just below / at / above the distance."""
    for u in (6, 13, 32, 50, 90):
        a, b = _edge_pairs(0xB175 + u, 70, 150, u)
        for k in (u - 1, u, u + 1, min(125, 2 * u)):
            got, plan = E.lev_bits(a, b, k, trans)
            assert got == oracle(a, b, k, trans), (u, k, trans, plan)


def test_bits_cfg2_cfg4_shapes():
    """BASELINE cfg2 (256 B, k = 32, Levenshtein) and cfg4 (128 B, k = 8, restricted Damerau) geometry."""
    ar, br = Dg.pairs_random(0x7A02, 40, 256)
    am, bm = Dg.pairs_mutated_fixed(0x7A12, 90, 256, 32)
    a = [x.tobytes() for x in ar] + [x.tobytes() for x in am]
    b = [x.tobytes() for x in br] + [x.tobytes() for x in bm]
    got, plan = E.lev_bits(a, b, 32, False)
    assert plan["s8"] and got == oracle(a, b, 32, False)
    assert sum(x is not None for x in got) >= 60
    ar, br = Dg.pairs_random(0x7A04, 40, 128)
    am, bm = Dg.pairs_mutated_fixed(0x7A14, 90, 128, 8, swaps=True)
    a = [x.tobytes() for x in ar] + [x.tobytes() for x in am]
    b = [x.tobytes() for x in br] + [x.tobytes() for x in bm]
    got, plan = E.lev_bits(a, b, 8, True)
    assert plan["NA"] == 3 and got == oracle(a, b, 8, True)
    assert sum(x is not None for x in got) >= 60


@pytest.mark.parametrize("chunk", [16, 32, 64])
def test_bits_long_strings_and_chunks(chunk):
    a, b = make_pairs(0xC0 + chunk, 64, 700, 40, True)
    for trans, k in [(False, 50), (True, 33), (False, 5)]:
        got, plan = E.lev_bits(a, b, k, trans, chunk=chunk)
        assert got == oracle(a, b, k, trans), (chunk, trans, k, plan)


def test_bits_null_bytes_and_degenerate():
    """Zero bytes are ordinary characters for Levenshtein (tests/basic_tests.rs:503-537); empty strings; 1-byte strings."""
    g = Dg.rng(77)
    a = [b"", b"", b"\0", b"\0\0\0", b"abc", b"\0a\0b", bytes(40), bytes(40), b"x", b"ab", b"ba"]
    b = [b"", b"\0\0", b"\0", b"\0", b"", b"a\0b\0", bytes(37), bytes([0] * 20 + [1] + [0] * 19), b"y", b"ba", b"ab"]
    for _ in range(60):
        x = bytes(g.integers(0, 2, size=int(g.integers(0, 30))).astype(np.uint8))
        y = bytes(g.integers(0, 2, size=int(g.integers(0, 30))).astype(np.uint8))
        a.append(x); b.append(y)
    for trans in (False, True):
        for k in (0, 1, 2, 5, 9, 40):
            got, plan = E.lev_bits(a, b, k, trans)
            assert got == oracle(a, b, k, trans), (k, trans, plan)


def test_bits_rejects_what_it_cannot_hold():
    with pytest.raises(RuntimeError):
        E.lev_bits([b"a" * 300], [b"b" * 300], 128, False)     # 129 diagonals > 128-bit window
    with pytest.raises(RuntimeError):
        E.lev_bits([b"a" * 300], [b"b" * 300], 126, True)      # 127 + 2 > 128
    got, _ = E.lev_bits([b"a" * 300], [b"b" * 300], 127, False)
    assert got == [None]


@pytest.mark.parametrize("trans", [False, True])
def test_bits_static_window_form(trans):
    """The static-window form (registers move a dword every 4th column, sub-column s reads bit i from byte i + s) against
    the oracle and against the sliding form: every window width it exists for, k at the edge of what the window holds,
    ragged lengths (groups of 4 columns cut by a pair's end), several chunks."""
    a, b = make_pairs(0x57A7, 130, 300, 20, trans)
    for na in (8, 9, 12, 16, 18, 24, 32):
        kmax = 4 * na - 3 - 1 - (2 if trans else 0)                     # widest band this window holds
        for k in sorted({0, 5, kmax - 1, kmax}):
            got, plan = E.lev_bits(a, b, k, trans, force_NA=na, static=2)
            assert plan["static"] and plan["NA"] == na
            assert got == oracle(a, b, k, trans), (na, k, trans, plan)
    ea, eb = _edge_pairs(0xB175 + 32, 70, 80, 32)
    for k in (31, 32, 33):
        got, plan = E.lev_bits(ea, eb, k, trans, static=2)
        slid, plan2 = E.lev_bits(ea, eb, k, trans, static=1)
        assert plan["static"] and not plan2["static"]
        assert got == slid == oracle(ea, eb, k, trans), (k, trans, plan, plan2)


@pytest.mark.parametrize("trans", [False, True])
def test_bits_stride8_window_form(trans):
    """The stride-8 form (33 diagonals in 8 registers, register m = the bytes of window bits m, m+8, m+16, m+24; renamed, not
    moved, from column to column; the 33rd diagonal as match | carry) against the oracle and the sliding form: every k it serves,
    ragged lengths (blocks of 8 columns cut by a pair's end, pairs ending inside a block while others run on), several chunks,
    pairs whose only optimal path runs along a band edge."""
    a, b = make_pairs(0x57A8, 140, 300, 20, trans)
    kmax = 33 - 1 - (2 if trans else 0)
    for k in sorted({0, 1, 5, 17, 24, kmax - 1, kmax}):
        got, plan = E.lev_bits(a, b, k, trans, static=3)
        assert plan["s8"] and plan["NA"] == 8
        assert got == oracle(a, b, k, trans), (k, trans, plan)
    for dist in (30, 32):
        ea, eb = _edge_pairs(0xB175 + dist, 70, 80, dist)
        for k in (kmax - 2, kmax - 1, kmax):
            got, plan = E.lev_bits(ea, eb, k, trans, static=3)
            slid, plan2 = E.lev_bits(ea, eb, k, trans, static=1)
            assert plan["s8"] and not plan2["s8"] and not plan2["static"]
            assert got == slid == oracle(ea, eb, k, trans), (dist, k, trans, plan, plan2)
    # short strings, empty strings, one side empty
    sa = [b"", b"a", b"", b"abc", b"kitten", b"x" * 40, b"ab" * 9]
    sb = [b"", b"", b"b", b"abd", b"sitting", b"x" * 9, b"ba" * 9]
    for k in (0, 3, kmax):
        got, plan = E.lev_bits(sa, sb, k, trans, static=3)
        assert plan["s8"] and got == oracle(sa, sb, k, trans), (k, trans)


def test_bits_planner_picks_the_static_form_from_8_dwords():
    got, plan = E.lev_bits([b"a" * 300], [b"b" * 300], 32, False)
    assert plan["s8"] and plan["NA"] == 8                                     # cfg2: 33 diagonals, the stride-8 form
    got, plan = E.lev_bits([b"a" * 300], [b"b" * 300], 20, False)
    assert not plan["s8"] and not plan["static"] and plan["NA"] == 6          # 21 diagonals: the sliding form is cheaper
    got, plan = E.lev_bits([b"a" * 300], [b"b" * 300], 8, True)
    assert not plan["static"] and plan["NA"] == 3                             # cfg4: too narrow to pay for itself
    got, plan = E.lev_bits([b"a" * 300], [b"b" * 300], 33, False)
    assert plan["static"] and plan["NA"] == 10


@pytest.mark.parametrize("trans", [False, True])
def test_bits_byte_values_around_the_perm_selector_codes(trans):
    """The byte test is a v_perm_b32 selector lookup (a ^ b ^ 0x0C == 12): alphabets of the selector codes that behave
    specially (0..13, 0x0C itself, 0x8C, 0xFF) must still compare exactly, sliding and static window forms."""
    g = np.random.default_rng(0xC0DE)
    for alpha in ([0x0C, 0x00], [0x00, 0x08, 0x0B, 0x0C, 0x0D], [0x0C, 0x8C, 0xFF, 0x04]):
        al = np.array(alpha, dtype=np.uint8)
        a = [al[g.integers(0, len(al), int(g.integers(0, 50)))].tobytes() for _ in range(64)]
        b = []
        for x in a:
            y = bytearray(x)
            for _ in range(int(g.integers(0, 6))):
                if y and g.random() < 0.5:
                    y[int(g.integers(0, len(y)))] = int(al[int(g.integers(0, len(al)))])
                elif y and g.random() < 0.5:
                    del y[int(g.integers(0, len(y)))]
                else:
                    y.insert(int(g.integers(0, len(y) + 1)), int(al[int(g.integers(0, len(al)))]))
            b.append(bytes(y))
        for k, na, static in ((12, 0, 0), (30, 9, 2)):
            got, plan = E.lev_bits(a, b, k, trans, force_NA=na, static=static)
            assert got == oracle(a, b, k, trans), (alpha[:4], k, na, static)


# ---- fixed-length (strided) batches take the LINE form: every 128-byte line of a string is fetched once, whole, and handed to
# LDS piece by piece (lev_bits_body.h)
def _fixed_batch(seed, n, la, lb, k, alpha=26, swaps=False):
    g = Dg.rng(seed)
    a = g.integers(97, 97 + alpha, size=(n, la), dtype=np.uint8)
    b = np.empty((n, lb), dtype=np.uint8)
    for i in range(n):
        if g.random() < 0.75:
            m = Dg.mutate(g, a[i].tobytes(), int(g.integers(0, k + 3)), swaps)
            m = (m + g.integers(97, 97 + alpha, size=lb, dtype=np.uint8).tobytes())[:lb]
        else:
            m = g.integers(97, 97 + alpha, size=lb, dtype=np.uint8).tobytes()
        b[i] = np.frombuffer(m, dtype=np.uint8)
    return a, b


@pytest.mark.parametrize("la,lb,k,trans", [(256, 256, 32, False), (128, 128, 8, True), (100, 93, 12, False), (61, 70, 20, True),
                                           (300, 300, 60, False), (17, 17, 3, False), (200, 215, 127, False), (1, 1, 1, False),
                                           (64, 64, 0, False), (130, 129, 33, True), (700, 690, 40, False), (513, 530, 19, True),
                                           (1000, 1000, 100, False), (128, 128, 5, False), (127, 129, 6, True), (16, 16, 16, False),
                                           (400, 385, 15, False), (400, 415, 17, True)])
def test_emu_bits_fixed_length_coalesced(la, lb, k, trans):
    """n = 150: two full wavefronts and one with 22 live lanes (helper lanes serve live pairs while their own pair is absent)."""
    a, b = _fixed_batch(la * 7 + lb + k, 150, la, lb, k, swaps=trans)
    costs = (1, 1, 0, 1 if trans else None)
    want = [O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), k, False, costs)[0] for i in range(150)]
    for static in (1, 2, 3):
        if static == 2 and k >= 124:
            continue
        if static == 3 and k + 1 + (2 if trans else 0) > 33:
            continue
        got, plan = E.lev_bits_fixed(a, b, k, trans, static=static)
        assert got == want, (la, lb, k, trans, static, plan)
    assert any(w is not None for w in want) or k == 0


@pytest.mark.parametrize("la,lb,k,trans", [(128, 128, 8, True), (128, 128, 5, False), (100, 93, 12, False), (61, 70, 20, True), (17, 17, 3, False),
                                           (1, 1, 1, False), (127, 120, 6, True), (64, 64, 0, False), (128, 128, 32, False), (96, 110, 30, True)])
def test_emu_bits_fixed_length_chunk_form(la, lb, k, trans):
    """Fixed-length batches of strings up to one line take the CHUNK form with wave-uniform load predicates (lanes without a pair
    read the batch's first pair): n = 150 leaves 42 lanes of the last wavefront without a pair."""
    a, b = _fixed_batch(la * 5 + lb + k, 150, la, lb, k, swaps=trans)
    costs = (1, 1, 0, 1 if trans else None)
    want = [O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), k, False, costs)[0] for i in range(150)]
    E.bits_fixed_chunk(True)
    try:
        for static in (1, 2, 3):
            if static == 3 and k + 1 + (2 if trans else 0) > 33:
                continue
            got, plan = E.lev_bits_fixed(a, b, k, trans, static=static)
            assert got == want, (la, lb, k, trans, static, plan)
    finally:
        E.bits_fixed_chunk(False)


def test_emu_bits_fixed_length_subset():
    """The levenshtein_exp rounds hand the kernel a subset list: helper lanes must follow the OWNER's pair index."""
    a, b = _fixed_batch(5, 200, 96, 96, 10)
    g = Dg.rng(9)
    subset = np.sort(g.choice(200, size=77, replace=False)).astype(np.uint32)
    got, _ = E.lev_bits_fixed(a, b, 10, False, subset=subset)
    for i in range(200):
        if i in set(int(x) for x in subset):
            assert got[i] == O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), 10, False, (1, 1, 0, None))[0], i
        else:
            assert got[i] == "untouched", i


def test_emu_bits_fixed_length_nul_and_high_bytes():
    g = Dg.rng(12)
    a = g.integers(0, 256, size=(70, 80), dtype=np.uint8)
    b = a.copy()
    b[:, ::7] = g.integers(0, 256, size=b[:, ::7].shape, dtype=np.uint8)
    a[:, 3] = 0; b[:, 5] = 12; a[:, 9] = 12
    got, _ = E.lev_bits_fixed(a, b, 30, False)
    want = [O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), 30, False, (1, 1, 0, None))[0] for i in range(70)]
    assert got == want


# ---- two pairs per lane (lev_bits2_body.h): narrow bands of fixed-length batches
@pytest.mark.parametrize("la,lb,k,trans", [(128, 128, 8, True), (128, 128, 8, False), (64, 64, 0, False), (64, 64, 1, True), (100, 97, 5, True),
                                           (100, 104, 12, False), (200, 200, 14, False), (200, 193, 12, True), (30, 30, 3, False),
                                           (17, 19, 2, True), (1, 1, 1, False), (300, 300, 4, True), (70, 70, 13, False), (129, 120, 10, True),
                                           (1000, 1000, 7, True), (513, 520, 10, False), (64, 64, 14, False), (65, 60, 12, True), (16, 16, 14, False),
                                           (2, 9, 12, True), (257, 255, 3, False)])
def test_emu_bits2_two_pairs_per_lane(la, lb, k, trans):
    """n = 300: two full 128-pair wavefronts and one with 44 pairs (pair B absent in most of its lanes)."""
    a, b = _fixed_batch(la * 11 + lb + k, 300, la, lb, k, swaps=trans)
    costs = (1, 1, 0, 1 if trans else None)
    want = [O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), k, False, costs)[0] for i in range(300)]
    got, plan = E.lev_bits2(a, b, k, trans)
    assert got is not None, "planner declined"
    assert got == want, (la, lb, k, trans, plan, [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w][:5])


def test_emu_bits2_declines_wide_bands_and_follows_subsets():
    a, b = _fixed_batch(3, 10, 64, 64, 20)
    assert E.lev_bits2(a, b, 15, False)[0] is None and E.lev_bits2(a, b, 13, True)[0] is None     # 16 / 16 diagonals
    assert E.lev_bits2(a, b, 14, False)[0] is not None and E.lev_bits2(a, b, 12, True)[0] is not None
    a, b = _fixed_batch(8, 400, 90, 90, 9)
    g = Dg.rng(4)
    subset = np.sort(g.choice(400, size=201, replace=False)).astype(np.uint32)
    got, _ = E.lev_bits2(a, b, 9, False, subset=subset)
    chosen = set(int(x) for x in subset)
    for i in range(400):
        if i in chosen:
            assert got[i] == O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), 9, False, (1, 1, 0, None))[0], i
        else:
            assert got[i] == "untouched", i


# ---- one pair, one wavefront, the recurrence on the scalar unit (lev_one_body.h)
@pytest.mark.parametrize("trans", [False, True])
def test_emu_lev_one_single_pair(trans):
    g = Dg.rng(77 + trans)
    costs = (1, 1, 0, 1 if trans else None)
    n_checked = 0
    for _ in range(400):
        la = int(g.choice([0, 1, 2, 5, 17, 63, 64, 65, 100, 256, 300, 700]))
        alpha = int(g.choice([2, 4, 26]))
        a = g.integers(97, 97 + alpha, size=la, dtype=np.uint8).tobytes()
        if g.random() < 0.7:
            b = Dg.mutate(g, a, int(g.integers(0, 40)), swaps=trans)
        else:
            b = g.integers(97, 97 + alpha, size=int(g.integers(0, la + 40)), dtype=np.uint8).tobytes()
        if not a and not b:
            continue
        k = int(g.choice([0, 1, 2, 7, 8, 15, 16, 30, 31, 32, 33, 40, 60, 61, 62, 63, 0xFFFFFFFF]))
        got = E.lev_one(a, b, k, trans)
        if got == "declined":
            continue
        want = O.levenshtein_naive_k_with_opts(a, b, k, False, costs)[0]
        assert got == want, (a, b, k, trans, got, want)
        n_checked += 1
    assert n_checked > 250


def test_emu_lev_one_edges():
    # the band limit: 64 diagonals (61 with the transposition rows); unbounded k on short strings clamps into it
    assert E.lev_one(b"a" * 100, b"b" * 100, 63, False) is None
    assert E.lev_one(b"a" * 100, b"b" * 100, 64, False) == "declined"
    assert E.lev_one(b"a" * 100, b"b" * 100, 61, True) is None and E.lev_one(b"a" * 100, b"b" * 100, 62, True) == "declined"
    assert E.lev_one(b"kitten", b"sitting", 0xFFFFFFFF, False) == 3
    assert E.lev_one(b"", b"abc", 5, False) == 3 and E.lev_one(b"abc", b"", 2, False) is None
    assert E.lev_one(b"abcdef", b"badcfe", 10, True) == 3 and E.lev_one(b"abcdef", b"badcfe", 10, False) == 4
    x = bytes(range(256)) * 4
    y = x[1:] + bytes([0])
    assert E.lev_one(x, x, 0, False) == 0 and E.lev_one(x, y, 2, False) == O.levenshtein_naive_k_with_opts(x, y, 2, False, LEV)[0] == 2


# ---- early out (LevParams::tune bit 1): same answers, the wavefront stops once none of its pairs can end at or below k
@pytest.mark.parametrize("la,lb,k,trans", [(256, 256, 32, False), (256, 250, 30, True), (400, 400, 25, False), (128, 128, 8, True),
                                           (128, 128, 8, False), (200, 207, 12, False), (300, 300, 4, True), (96, 96, 0, False)])
def test_emu_early_out_same_answers(la, lb, k, trans):
    """Waves of far pairs only (they stop early), of near pairs only, and mixed ones (one near pair keeps its wavefront going)."""
    g = Dg.rng(la + lb + k)
    n = 64 * 5 + 17
    a, b = _fixed_batch(la * 7 + k, n, la, lb, k, swaps=trans)                 # near pairs ...
    far = np.zeros(n, dtype=bool)
    far[:128] = True                                                           # ... two wavefronts of random (far) pairs ...
    far[192:256] = g.random(64) < 0.9                                          # ... and a mixed one
    b[far] = g.integers(33, 127, size=(int(far.sum()), lb), dtype=np.uint8)
    costs = (1, 1, 0, 1 if trans else None)
    want = [O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), k, False, costs)[0] for i in range(n)]
    try:
        E.bits_set_tune(2)
        got, plan = E.lev_bits_fixed(a, b, k, trans)
        assert got == want, (plan, [(i, x, y) for i, (x, y) in enumerate(zip(got, want)) if x != y][:5])
        got2, _ = E.lev_bits2(a, b, k, trans)
        if got2 is not None:
            assert got2 == want
    finally:
        E.bits_set_tune(0)
    assert any(w is None for w in want[:128]) and any(w is not None for w in want[256:])


# ---- small alphabets (lev_bitsq_body.h): the match vector from per-symbol tables
def _dna_batch(seed, n, la, lb, k, alphabet, swaps=False):
    g = Dg.rng(seed)
    sym = np.frombuffer(bytes(alphabet), dtype=np.uint8)
    a = sym[g.integers(0, len(sym), size=(n, la))]
    b = np.empty((n, lb), dtype=np.uint8)
    for i in range(n):
        r = i % 4
        if r == 0:
            b[i] = sym[g.integers(0, len(sym), size=lb)]                       # unrelated
        else:
            s = bytearray(a[i].tobytes())
            for _ in range(int(g.integers(0, k + 2))):
                t = int(g.integers(0, 4 if swaps else 3))
                p = int(g.integers(0, max(1, len(s))))
                if t == 0 and s: s[p] = int(sym[g.integers(0, len(sym))])
                elif t == 1: s.insert(p, int(sym[g.integers(0, len(sym))]))
                elif t == 2 and s: del s[p]
                elif t == 3 and len(s) > 1 and p + 1 < len(s): s[p], s[p + 1] = s[p + 1], s[p]
            s = (bytes(s) + sym[g.integers(0, len(sym), size=lb)].tobytes())[:lb]
            b[i] = np.frombuffer(s, dtype=np.uint8)
    return a, b


@pytest.mark.parametrize("la,lb,k,trans,alphabet", [
    (256, 256, 32, False, b"ACGT"), (256, 256, 30, True, b"ACGT"), (128, 128, 8, True, b"acgt"), (100, 97, 12, False, b"ACGU"),
    (100, 110, 20, True, b"ACGT"), (300, 300, 31, False, b"TGCA"), (513, 520, 25, False, b"ACGT"), (64, 64, 0, False, b"AC"),
    (40, 40, 5, True, bytes([0, 1, 2, 3])), (17, 30, 14, False, b"ACG"), (1, 1, 1, False, b"A"), (1000, 1000, 7, True, b"ACGT"),
    (255, 257, 32, False, b"ACGT"), (129, 160, 32, False, b"ACGT"), (160, 129, 30, True, b"ACGT")])
def test_emu_bitsq_small_alphabets(la, lb, k, trans, alphabet):
    n = 64 * 3 + 9
    a, b = _dna_batch(la * 5 + lb + k, n, la, lb, k, alphabet, swaps=trans)
    costs = (1, 1, 0, 1 if trans else None)
    want = [O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), k, False, costs)[0] for i in range(n)]
    got, bad = E.lev_bitsq(a, b, k, alphabet, trans)
    assert got is not None, "declined"
    assert bad == [] and got == want, (bad[:5], [(i, x, y) for i, (x, y) in enumerate(zip(got, want)) if x != y][:5])


def test_emu_bitsq_foreign_bytes_and_subsets():
    """Pairs with a byte outside the alphabet -- anywhere in either string, the last byte included -- are left to the caller's
    fallback (listed, out untouched); the others are answered.  Alphabets without a code hash and wide bands are declined."""
    a, b = _dna_batch(7, 200, 150, 150, 20, b"ACGT")
    foreign = {3: (0, 0), 64: (1, 149), 65: (0, 77), 130: (1, 16), 199: (0, 149)}
    for i, (which, pos) in foreign.items():
        (a if which == 0 else b)[i, pos] = ord("N")
    got, bad = E.lev_bitsq(a, b, 20, b"ACGT")
    assert bad == sorted(foreign)
    for i in range(200):
        if i in foreign:
            assert got[i] == "untouched"
        else:
            assert got[i] == O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), 20, False, (1, 1, 0, None))[0], i
    assert E.lev_bitsq(a, b, 20, b"@AQP")[0] is None            # 0x40 0x41 0x51 0x50: no two adjacent bits tell the four apart
    assert E.lev_bitsq(a, b, 33, b"ACGT")[0] is None and E.lev_bitsq(a, b, 31, b"ACGT", trans=True)[0] is None
    sub = np.arange(0, 200, 3, dtype=np.uint32)
    got, bad = E.lev_bitsq(a, b, 20, b"ACGT", subset=sub)
    assert bad == [i for i in sorted(foreign) if i % 3 == 0]
    for i in range(200):
        assert (got[i] == "untouched") == (i % 3 != 0 or i in foreign)


# ---- alphabets of up to 32 symbols (lev_bitsqw_body.h): dense rings of 64 rows, `b` looked up by the byte itself
PROTEIN = b"ACDEFGHIKLMNPQRSTVWY"
IUPAC = b"ACGTRYSWKMBDHVNU"


@pytest.mark.parametrize("la,lb,k,trans,alphabet", [
    (256, 256, 32, False, PROTEIN), (256, 256, 30, True, PROTEIN), (128, 128, 8, True, IUPAC), (100, 97, 12, False, IUPAC),
    (100, 110, 20, True, PROTEIN), (300, 300, 31, False, b"0123456789"), (513, 520, 25, False, PROTEIN), (64, 64, 0, False, b"AC"),
    (40, 40, 5, True, bytes(range(32))), (17, 30, 14, False, b"abcdefghijklmnopqrstuvwxyz"), (1, 1, 1, False, b"A"),
    (1000, 1000, 7, True, IUPAC), (255, 257, 32, False, PROTEIN), (129, 160, 32, False, IUPAC), (160, 129, 30, True, PROTEIN),
    (290, 258, 32, False, PROTEIN), (200, 168, 32, False, b"ACGT"), (80, 50, 30, True, bytes(range(0x40, 0x60)))])
def test_emu_bitsqw_alphabets_up_to_32(la, lb, k, trans, alphabet):
    n = 64 * 2 + 9
    a, b = _dna_batch(la * 7 + lb + k, n, la, lb, k, alphabet, swaps=trans)
    costs = (1, 1, 0, 1 if trans else None)
    want = [O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), k, False, costs)[0] for i in range(n)]
    got, bad = E.lev_bitsqw(a, b, k, alphabet, trans)
    assert got is not None, "declined"
    assert bad == [] and got == want, (bad[:5], [(i, x, y) for i, (x, y) in enumerate(zip(got, want)) if x != y][:5])


def test_emu_bitsqw_foreign_bytes_and_subsets():
    """A byte outside the alphabet -- a code no symbol has ('B', 'X'), a symbol's code under other high bits ('a', 0xC1), in either string,
    first / middle / last position -- sends the pair to the fallback list; the others are answered.  Alphabets without a 5-bit code whose
    other bits agree are declined."""
    a, b = _dna_batch(11, 200, 150, 150, 20, PROTEIN)
    foreign = {3: (0, 0, ord("B")), 64: (1, 149, ord("X")), 65: (0, 77, ord("a")), 130: (1, 16, 0xC1), 199: (0, 149, ord("Z")), 100: (1, 0, 0),
               7: (0, 15, ord("c")), 8: (0, 16, ord("J"))}
    for i, (which, pos, byte) in foreign.items():
        (a if which == 0 else b)[i, pos] = byte
    got, bad = E.lev_bitsqw(a, b, 20, PROTEIN)
    assert bad == sorted(foreign)
    for i in range(200):
        if i in foreign:
            assert got[i] == "untouched"
        else:
            assert got[i] == O.levenshtein_naive_k_with_opts(a[i].tobytes(), b[i].tobytes(), 20, False, (1, 1, 0, None))[0], i
    assert E.lev_bitsqw(a, b, 20, b"ACGTacgt")[0] is None          # two cases: the codes collide at every shift a byte's other bits allow
    assert E.lev_bitsqw(a, b, 20, b"AB0")[0] is None               # distinct codes, other bits differ
    assert E.lev_bitsqw(a, b, 33, PROTEIN)[0] is None and E.lev_bitsqw(a, b, 31, PROTEIN, trans=True)[0] is None
    sub = np.arange(0, 200, 3, dtype=np.uint32)
    got, bad = E.lev_bitsqw(a, b, 20, PROTEIN, subset=sub)
    assert bad == [i for i in sorted(foreign) if i % 3 == 0]
    for i in range(200):
        assert (got[i] == "untouched") == (i % 3 != 0 or i in foreign)
