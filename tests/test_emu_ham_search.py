"""not-gpu: the shift-add hamming_search scan (ham_search_body.h) against the oracle's scalar hamming_search."""
import numpy as np
import pytest

import datagen as Dg
import emu_lib as E
import oracle_lib as O


def test_shift_add_equals_oracle():
    g = Dg.rng(0x4A)
    for n in (1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 31, 32):
        needle = bytes(g.integers(1, 256, size=n).astype(np.uint8))
        hay = bytearray(g.integers(1, 256, size=3000).astype(np.uint8).tobytes())
        for pos in range(50, 2900, 211):                                      # planted copies with a few substitutions
            m = bytearray(needle)
            for _ in range(int(g.integers(0, 4))):
                m[int(g.integers(0, n))] = int(g.integers(1, 256))
            hay[pos:pos + n] = m
        hay = bytes(hay)
        for k in sorted({0, 1, n // 2, n}):
            want = O.hamming_search_naive_with_opts(needle, hay, k, O.ALL)
            for tile in (64, 1000):
                assert E.ham_search(needle, hay, k, tile=tile) == want, (n, k, tile)
            assert E.ham_search(needle, hay, k, tile=500, words=8) == want


def test_shift_add_small_alphabet_and_edges():
    g = Dg.rng(0x4B)
    for n in (3, 7, 20, 32):
        needle = bytes(g.integers(97, 99, size=n).astype(np.uint8))
        hay = bytes(g.integers(97, 99, size=400).astype(np.uint8))
        for k in (0, n // 3, n - 1):
            assert E.ham_search(needle, hay, k, tile=96) == O.hamming_search_naive_with_opts(needle, hay, k, O.ALL)
    assert E.ham_search(b"abc", b"abc", 0) == [(0, 3, 0)]
    assert E.ham_search(b"abc", b"abd", 1) == [(0, 3, 1)]


def test_swar16_equals_oracle():
    """The SWAR form (one lane per 16 offsets, windows shifted once and shared; ham_swar_body.h): every needle length 1..32 and some up to 64, every start
    alignment of the haystack, k from 0 to n; NUL bytes in the needle's padding positions must not match haystack zeros."""
    g = Dg.rng(0x4C)
    for n in list(range(1, 33)) + [33, 40, 47, 48, 49, 63, 64]:
        needle = bytes(g.integers(1, 256, size=n).astype(np.uint8))
        hay = bytearray(g.integers(0, 256, size=700).astype(np.uint8).tobytes())      # zeros included: the kernel counts, the contract check is separate
        for pos in range(5, 650, 53):
            m = bytearray(needle)
            for _ in range(int(g.integers(0, 4))):
                m[int(g.integers(0, n))] = int(g.integers(0, 256))
            hay[pos:pos + n] = m
        hay = bytes(hay)
        for k in sorted({0, 1, n // 2, n}):
            want = O.hamming_search_naive_with_opts(needle, hay, k, O.ALL)
            for delta in (0, 1, 7, 15) if n % 5 else range(16):
                assert sorted(E.ham_search_swar(needle, hay, k, delta)) == want, (n, k, delta)
    assert E.ham_search_swar(b"abc", b"abc", 0) == [(0, 3, 0)]
    assert E.ham_search_swar(b"abc", b"xxabd", 1, 3) == [(2, 5, 1)]


def test_bit_sliced_equals_oracle():
    """Bit-sliced mismatch counters (ham_bits_body.h): needle lengths 1..32, every k below the needle length that fits five counter bits
    (the bias 2^B - 1 - k turns "more than k" into a counter overflow), tiles that cut planted copies, haystacks with zeros."""
    g = Dg.rng(0x4D)
    for n in (1, 2, 5, 8, 9, 12, 16, 17, 24, 31, 32):
        needle = bytes(g.integers(0, 256, size=n).astype(np.uint8))
        hay = bytearray(g.integers(0, 256, size=1500).astype(np.uint8).tobytes())
        for pos in range(3, 1450, 61):
            m = bytearray(needle)
            for _ in range(int(g.integers(0, 6))):
                m[int(g.integers(0, n))] = int(g.integers(0, 256))
            hay[pos:pos + n] = m
        hay = bytes(hay)
        for k in sorted({0, 1, 2, 3, 4, 7, 8, 14, 15, 16, 30, 31}):
            if k >= n:
                continue
            want = O.hamming_search_naive_with_opts(needle, hay, k, O.ALL)
            for tile in (128, 256, 640):
                got = E.ham_search_bits(needle, hay, k, tile)
                assert got is not None and sorted(got) == want, (n, k, tile)
    assert E.ham_search_bits(b"abcd", b"xxabcdxx", 4) is None and E.ham_search_bits(b"abcd" * 8, b"q" * 100, 31) == []


def test_phase_plan_rules():
    """(needle length, k) -> (phases per dword, positions counted, counter bits): the subset stays selective (L >= 2k, L - k >= 4) unless it
    is the whole needle; the BASELINE-style rows land where DESIGN says."""
    pl = lambda n, k: (E.ham_search_phase(bytes(range(1, n + 1)), bytes(300), k) or (None, None))[1]
    assert pl(32, 8) == (2, 16, 4) and pl(64, 16) == (1, 32, 5) and pl(32, 2) == (4, 8, 2) and pl(16, 2) == (2, 8, 2)
    assert pl(64, 17) is None and pl(32, 20) == (1, 32, 5) and pl(32, 32) is None and pl(200, 3) == (4, 8, 2) and pl(200, 2) == (4, 8, 2) and pl(8, 2) == (1, 8, 2)
    assert pl(16, 4) == (2, 8, 3) and pl(24, 6) == (2, 12, 3)          # (round 6: a 16-byte needle takes two phases of 8, not one of 16)
    for n in range(1, 80):
        for k in range(0, 32):
            p = pl(n, k)
            if p is None:
                continue
            q, l, b = p
            assert k < n and (1 << b) - 1 >= k and q * l <= 32 and q * (l - 1) <= n - 1
            assert (q == 1 and l == n) or (l >= 2 * k and l - k >= 4), (n, k, p)


def test_phase_form_equals_oracle():
    """The phased bit-sliced filter (ham_phase_body.h): every needle length 1..72 and some longer, every k the plan takes up to n/2, tiles
    that cut planted copies, every forced phase count, haystack lengths that are no multiple of the phase count, zeros in the haystack."""
    g = Dg.rng(0x4E)
    for n in list(range(1, 73)) + [100, 127, 128, 200]:
        needle = bytes(g.integers(0, 256, size=n).astype(np.uint8))
        hl = 1400 + int(g.integers(0, 4))
        hay = bytearray(g.integers(0, 256, size=hl).astype(np.uint8).tobytes())
        for pos in range(3, hl - n - 1, 97):
            m = bytearray(needle)
            for _ in range(int(g.integers(0, max(2, n // 2 + 2)))):
                m[int(g.integers(0, n))] = int(g.integers(0, 256))
            hay[pos:pos + n] = m
        hay[hl - n:] = needle                                                  # a hit that ends with the haystack
        hay = bytes(hay)
        for k in sorted({0, 1, 2, 3, 5, 8, n // 4, n // 3, n // 2, min(31, n - 1)}):
            if k >= n:
                continue
            want = O.hamming_search_naive_with_opts(needle, hay, k, O.ALL)
            for tile, qf in ((128, 0), (256, 1), (640, 2), (384, 0)):
                r = E.ham_search_phase(needle, hay, k, tile, qf)
                if r is None:
                    continue
                got, plan, cand = r
                assert sorted(got) == want, (n, k, tile, qf, plan)
                assert cand >= len(want)


def test_phase_form_low_entropy_and_dense_hits():
    """Four-letter and one-letter texts: the subset filter passes many candidates, the recount decides; dense hits (every offset)."""
    g = Dg.rng(0x4F)
    for n, k in ((32, 8), (64, 16), (33, 2), (16, 2), (40, 10), (12, 0)):
        needle = bytes(g.integers(97, 101, size=n).astype(np.uint8))
        hay = bytes(g.integers(97, 101, size=900).astype(np.uint8))
        for tile in (128, 512):
            got, plan, cand = E.ham_search_phase(needle, hay, k, tile)
            assert sorted(got) == O.hamming_search_naive_with_opts(needle, hay, k, O.ALL), (n, k, plan)
    got, plan, cand = E.ham_search_phase(b"a" * 40, b"a" * 500, 3, 128)
    assert sorted(got) == [(p, p + 40, 0) for p in range(461)] and cand == 461
