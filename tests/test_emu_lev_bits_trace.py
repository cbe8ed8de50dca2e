"""not-gpu: batch tracebacks by checkpoints + recomputation (lev_bits_trace_body.h) in the 64-lane host emulation against the oracle's
scripts (src/levenshtein.rs:493-532 argmin order, :561-606 walk), edit for edit."""
import numpy as np
import pytest

import datagen as Dg
import emu_lib as E
import oracle_lib as O


def _check(a, b, k, trans, tile, fixed=False):
    costs = (1, 1, 0, 1) if trans else (1, 1, 0, None)
    want = [O.levenshtein_naive_k_with_opts(x, y, k, True, costs) for x, y in zip(a, b)]
    dists = [w[0] for w in want]
    max_len = max([len(x) for x in a] + [len(y) for y in b] + [1])
    u = min(k, max_len)
    got = E.lev_bits_trace(a, b, u, dists, trans, tile, fixed)
    for p, (g, w) in enumerate(zip(got, want)):
        assert g == w[1], (p, a[p], b[p], k, trans, tile, g, w)
    return sum(1 for d in dists if d is not None)


@pytest.mark.parametrize("trans", [False, True])
@pytest.mark.parametrize("tile", [8, 16, 32])
def test_trace_mutated_ragged(trans, tile):
    g = Dg.rng(0x7B1 + tile + int(trans))
    for k in (3, 8, 17, 30, 32):
        if trans and k > 30:
            continue
        a, b = [], []
        for i in range(150):
            n = int(g.integers(0, 140))
            x = Dg.rand_str(g, n)
            y = Dg.mutate(g, x, int(g.integers(0, k + 3)), trans) if i % 5 else Dg.rand_str(g, int(g.integers(0, 140)))
            if i % 2:
                x, y = y, x                                              # both orientations: the kernel swaps the shorter onto the rows
            a.append(x); b.append(y)
        assert _check(a, b, k, trans, tile) > 60


@pytest.mark.parametrize("trans", [False, True])
def test_trace_small_alphabet_ties(trans):
    """Binary strings: many equal-cost paths -- the tie order decides the script."""
    g = Dg.rng(0x7B9 + int(trans))
    for k in (2, 6, 12, 30):
        a, b = [], []
        for i in range(130):
            n = int(g.integers(1, 90))
            x = bytes(g.integers(97, 99, size=n).astype(np.uint8))
            y = bytearray(x)
            for _ in range(int(g.integers(0, k + 1))):
                t = int(g.integers(0, 3))
                pos = int(g.integers(0, len(y) + 1))
                if t == 0 and pos < len(y):
                    y[pos] = 97 + int(g.integers(0, 2))
                elif t == 1:
                    y.insert(pos, 97 + int(g.integers(0, 2)))
                elif pos < len(y):
                    del y[pos]
            a.append(x); b.append(bytes(y))
        assert _check(a, b, k, trans, 16) > 40
        _check(a, b, k, trans, 8)


@pytest.mark.parametrize("trans", [False, True])
def test_trace_fixed_length_and_long(trans):
    """Fixed-length (strided) batches, strings over several tiles and lines, empty strings, identical pairs, None pairs in the wavefront."""
    g = Dg.rng(0x7C3 + int(trans))
    am, bm = Dg.pairs_mutated_fixed(0x7C4, 100, 256, 12, swaps=trans)
    a = [bytes(r) for r in am]; b = [bytes(r) for r in bm]
    b[7] = a[7]
    b[9] = bytes(g.integers(1, 255, size=256).astype(np.uint8))
    assert _check(a, b, 30 if trans else 32, trans, 16, fixed=True) > 90
    assert _check(a, b, 30 if trans else 32, trans, 32, fixed=True) > 90
    # tile 116: the forward sweep done by the distance kernel's CKPT instantiation (the launcher's route for fixed-length batches), line form
    assert _check(a, b, 30 if trans else 32, trans, 116, fixed=True) > 90
    for la, lb, k in ((100, 96, 9), (96, 100, 9), (128, 128, 20), (40, 47, 3), (300, 290, 30), (17, 17, 32 if not trans else 30), (16, 32, 16), (1, 1, 2)):
        am2, bm2 = Dg.pairs_mutated_fixed(0x7C5 + la + lb, 70, max(la, lb), max(1, k // 2), swaps=trans)
        a3 = [bytes(r[:la]) for r in am2]; b3 = [bytes(r[:lb]) for r in bm2]
        b3[3] = bytes(g.integers(1, 255, size=lb).astype(np.uint8))
        _check(a3, b3, k, trans, 116, fixed=True)
    assert _check(a, b, 30 if trans else 32, trans, 8, fixed=True) > 90
    a2 = [b"", b"abc", b"", b"kitten", b"ab", b"ba", b"abcdefgh" * 40, b"x" * 300]
    b2 = [b"", b"", b"xyz", b"sitting", b"ba", b"ab", (b"abcdefgh" * 40)[3:] + b"zz", b"x" * 290 + b"y" * 5]
    _check(a2, b2, 20, trans, 16)
    _check(a2, b2, 20, trans, 8)
    _check(a2, b2, 20, trans, 32)


@pytest.mark.parametrize("trans", [False, True])
def test_trace_folded_sweep_on_csr_batches(trans):
    """Tile 116 on CSR batches: the distance kernel's CKPT instantiation (chunk form, rows = the shorter string pair by pair) leaves the
    checkpoints, the trace kernel's HAVE_CKPT instantiation starts with the walk -- ragged lengths inside a wavefront (capped blocks), both
    orientations, None pairs, pairs outside the band, empty strings, strings over several chunks."""
    g = Dg.rng(0x7D1 + int(trans))
    for k in (3, 12, 30 if trans else 32):
        a, b = [], []
        for i in range(150):
            n = int(g.integers(0, 330 if i % 11 == 0 else 140))
            x = Dg.rand_str(g, n)
            y = Dg.mutate(g, x, int(g.integers(0, k + 3)), trans) if i % 5 else Dg.rand_str(g, int(g.integers(0, 140)))
            if i % 2:
                x, y = y, x
            a.append(x); b.append(y)
        a[5], b[5] = b"", b""
        a[6], b[6] = b"", b"ab"
        assert _check(a, b, k, trans, 116) > 60


# ---- adversarial alphabets: NUL is what load_strings pads with outside the strings (lev_bits_trace_body.h: pieces
# outside a string are zeroed) and 0x0C is the byte-test constant (`x ^ 0x0C0C0C0C`, wave.h ne12): strings made of exactly those bytes
# must trace like any other (the reference's own NUL cases: tests/basic_tests.rs:503-537, 774-802)
ALPHABETS = {"nul": [0], "nul_0c": [0, 0x0C], "0c_0d": [0x0C, 0x0D], "all": list(range(256))}


def _alpha_batch(g, sym, n_pairs, max_len, k, trans):
    sym = np.array(sym, dtype=np.uint8)
    a, b = [], []
    for i in range(n_pairs):
        n = int(g.integers(0, max_len))
        x = bytearray(sym[g.integers(0, len(sym), size=n)].tobytes())
        y = bytearray(x)
        for _ in range(int(g.integers(0, k + 2))):
            t = int(g.integers(0, 4 if trans else 3))
            pos = int(g.integers(0, len(y) + 1))
            c = int(sym[int(g.integers(0, len(sym)))])
            if t == 0 and pos < len(y):
                y[pos] = c
            elif t == 1:
                y.insert(pos, c)
            elif t == 2 and pos < len(y):
                del y[pos]
            elif t == 3 and pos + 1 < len(y):
                y[pos], y[pos + 1] = y[pos + 1], y[pos]
        x, y = bytes(x), bytes(y)
        if i % 2:
            x, y = y, x
        a.append(x); b.append(y)
    return a, b


@pytest.mark.parametrize("name", sorted(ALPHABETS))
@pytest.mark.parametrize("trans", [False, True])
@pytest.mark.parametrize("tile", [8, 16, 32, 116])
def test_trace_adversarial_alphabets(name, trans, tile):
    g = Dg.rng(0xAD5 + tile + 7 * int(trans) + len(name))
    for k in (4, 13, 30):
        if tile == 116:                                    # the folded sweep: fixed-length batches
            sym = np.array(ALPHABETS[name], dtype=np.uint8)
            L = 150
            a = [bytes(sym[g.integers(0, len(sym), size=L)]) for _ in range(70)]
            b = []
            for x in a:
                y = bytearray(x)
                for q in g.integers(0, L, size=int(g.integers(0, k // 2 + 1))):
                    y[int(q)] = int(sym[int(g.integers(0, len(sym)))])
                if trans and len(sym) > 1:
                    q = int(g.integers(0, L - 1))
                    y[q], y[q + 1] = y[q + 1], y[q]
                b.append(bytes(y))
            assert _check(a, b, k, trans, tile, fixed=True) > 30
        else:
            a, b = _alpha_batch(g, ALPHABETS[name], 110, 150, k, trans)
            assert _check(a, b, k, trans, tile) > 40


@pytest.mark.parametrize("trans", [False, True])
def test_trace_packed_form(trans):
    """LevBitsTraceParams::packed_cap: the walk writes each run where it belongs -- right-aligned in the pair's slot, front to back -- and a
    script of more runs than the slot holds keeps its last runs; n_runs still says how long the script is."""
    g = Dg.rng(0x9AC + int(trans))
    costs = (1, 1, 0, 1) if trans else (1, 1, 0, None)
    for k, cap in ((6, 13), (20, 41), (20, 7), (30, 61)):
        a, b = _alpha_batch(g, list(range(97, 101)), 100, 120, k, trans)
        want = [O.levenshtein_naive_k_with_opts(x, y, k, True, costs) for x, y in zip(a, b)]
        dists = [w[0] for w in want]
        u = min(k, max([len(x) for x in a] + [len(y) for y in b] + [1]))
        for tile in (16, 116) if False else (16,):
            got = E.lev_bits_trace(a, b, u, dists, trans, tile, False, packed=cap)
            cut = 0
            for p, (gp, w) in enumerate(zip(got, want)):
                if w[0] is None:
                    assert gp is None
                    continue
                nr, script = gp
                assert nr == len(w[1]) and script == w[1][max(0, len(w[1]) - cap):], (p, k, cap)
                cut += nr > cap
            assert cap >= 2 * k + 1 or cut > 0
