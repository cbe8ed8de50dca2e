"""-m gpu: levenshtein_search / hamming_search kernels (through the C ABI) against the scalar oracle."""
import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O

pytestmark = pytest.mark.gpu

COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (3, 1, 0, None), (1, 1, 2, None), (2, 1, 2, None), (2, 2, 1, 3)]


def prod_search(needle, hay, k, st, costs, anchored=False):
    import triple_accel_amd as T
    return [tuple(m) for m in T.levenshtein_search_simd_with_opts(needle, hay, k, st, T.EditCosts(*costs), anchored)]


@pytest.mark.parametrize("costs", COSTS)
def test_search_all_and_best_vs_oracle(costs):
    g = Dg.rng(41)
    for n in (1, 3, 8, 13, 24, 32):
        needle = Dg.rand_str(g, n)
        for k in (0, 1, (n + 1) // 2):
            hay = Dg.planted_haystack(int(g.integers(1 << 30)), needle, 20000, 150, max(1, k))
            for st in (O.ALL, O.BEST):
                want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, False)
                assert prod_search(needle, hay, k, st, costs) == want, (n, k, st, costs)


def test_search_ties_small_alphabet():
    g = Dg.rng(8)
    for costs in COSTS:
        for _ in range(15):
            n = int(g.integers(1, 9))
            needle = g.integers(97, 99, size=n, dtype=np.uint8).tobytes()
            hay = g.integers(97, 99, size=int(g.integers(0, 300)), dtype=np.uint8).tobytes()
            k = int(g.integers(0, n + 1))
            for st in (O.ALL, O.BEST):
                for anchored in (False, True):
                    want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, anchored)
                    assert prod_search(needle, hay, k, st, costs, anchored) == want, (needle, hay, k, st, anchored)


def test_default_search_is_best_half_k():
    import triple_accel_amd as T
    needle = b"helllo world!!"
    hay = Dg.planted_haystack(5, needle, 5000, 400, 3)
    want = O.levenshtein_search_naive_with_opts(needle, hay, O.default_search_k(len(needle)), O.BEST)
    assert [tuple(m) for m in T.levenshtein_search(needle, hay)] == want
    assert [tuple(m) for m in T.levenshtein_search(b"", b"abc")] == []


def test_search_cfg5_geometry_shard():
    """BASELINE cfg5 geometry scaled to what the oracle finishes in seconds: 32 B needle, k = 16, Best,
    a 4 MiB random shard with planted mutated copies."""
    import triple_accel_amd as T
    g = Dg.rng(0x7A05)
    needle = Dg.random_bytes(g, 32).tobytes()
    hay = bytearray(Dg.random_bytes(g, 4 << 20).tobytes())
    for pos in range(10000, len(hay) - 100, 300000):
        m = Dg.mutate(g, needle, 10)
        hay[pos:pos + len(m)] = m
    hay = bytes(hay)
    for st in (O.ALL, O.BEST):
        want = O.levenshtein_search_naive_with_opts(needle, hay, 16, st)
        got = [tuple(m) for m in T.levenshtein_search_simd_with_opts(needle, hay, 16, st, T.LEVENSHTEIN_COSTS, False)]
        assert got == want and len(want) > 0


def test_hamming_search():
    import triple_accel_amd as T
    g = Dg.rng(9)
    for n in (1, 3, 4, 10, 33, 100):
        needle = Dg.rand_str(g, n)
        hay = bytearray(Dg.rand_str(g, 30000))
        for pos in range(100, 29000, 777):
            m = bytearray(needle)
            for p in g.choice(n, size=min(n, 2), replace=False):
                m[p] = 32
            hay[pos:pos + n] = m
        hay = bytes(hay)
        for k in (0, 1, 3):
            for st in (O.ALL, O.BEST):
                want = O.hamming_search_simd_with_opts(needle, hay, k, st)
                got = [tuple(m) for m in T.hamming_search_simd_with_opts(needle, hay, k, st)]
                assert got == want, (n, k, st)
    assert list(T.hamming_search(b"abc", b"ab")) == []
    assert list(T.hamming_search(b"", b"ab")) == []
    with pytest.raises(T.PanicError):
        list(T.hamming_search(b"ab", b"a\x00b"))
    want = O.hamming_search_simd_with_opts(b"abc", b"  abc  abb", O.default_search_k(3), O.BEST)
    assert [tuple(m) for m in T.hamming_search(b"abc", b"  abc  abb")] == want


def test_long_needles_and_long_hamming_needles():
    """levenshtein_search with needles > 32 B (memory-backed column kernel) and hamming_search with needles > 256 B."""
    import triple_accel_amd as T
    g = Dg.rng(177)
    for costs in [(1, 1, 0, None), (1, 1, 0, 1), (2, 1, 2, None)]:
        for n in (33, 64, 150, 400):
            needle = Dg.rand_str(g, n)
            k = n // 6
            hay = Dg.planted_haystack(int(g.integers(1 << 30)), needle, 30000, 5000, max(1, k))
            for st in (O.ALL, O.BEST):
                want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, False)
                assert prod_search(needle, hay, k, st, costs) == want, (n, st, costs)
    needle = Dg.rand_str(g, 700)
    hay = bytearray(Dg.rand_str(g, 20000))
    hay[5000:5700] = needle
    hay[9000:9700] = needle[:350] + b"!" + needle[351:]
    hay = bytes(hay)
    for k in (0, 1, 5):
        want = O.hamming_search_simd_with_opts(needle, hay, k, O.ALL)
        assert [tuple(m) for m in T.hamming_search_simd_with_opts(needle, hay, k, O.ALL)] == want and len(want) >= 1


def test_unpacked_register_kernel_still_agrees(monkeypatch):
    """TA_SEARCH_UNPACKED=1 selects the 32-bit cost / 32-bit length register kernel (used when costs or tiles do
    not fit the packed 16+16 form); it must give the same matches."""
    monkeypatch.setenv("TA_SEARCH_UNPACKED", "1")
    g = Dg.rng(5)
    for costs in COSTS[:4]:
        needle = Dg.rand_str(g, 20)
        hay = Dg.planted_haystack(11, needle, 20000, 300, 6)
        for st in (O.ALL, O.BEST):
            assert prod_search(needle, hay, 6, st, costs) == O.levenshtein_search_naive_with_opts(needle, hay, 6, st, costs, False)


def test_big_k_takes_the_unpacked_kernel():
    g = Dg.rng(6)
    needle = Dg.rand_str(g, 10)
    hay = Dg.rand_str(g, 3000)
    costs = (255, 255, 255, None)
    for k in (40000, 70000):
        want = O.levenshtein_search_naive_with_opts(needle, hay, k, O.ALL, costs, False)
        assert prod_search(needle, hay, k, O.ALL, costs) == want


def test_hamming_search_alignment_and_edges():
    """Every haystack alignment (interior pointers), needle lengths around the 4-byte word size, offsets at both ends."""
    import torch
    from triple_accel_amd import batch as B
    g = Dg.rng(91)
    base = Dg.rand_str(g, 5000)
    big = torch.zeros(5000 + 64, dtype=torch.uint8, device="cuda")
    for shift in range(0, 8):
        big.zero_()
        big[shift:shift + 5000] = torch.frombuffer(bytearray(base), dtype=torch.uint8).cuda()
        view = big[shift:]
        for n in (1, 2, 3, 4, 5, 7, 8, 9, 31, 33):
            needle = base[100:100 + n]
            for hl in (5000, 4999, 4997):
                want = O.hamming_search_naive_with_opts(needle, base[:hl], n // 3, O.ALL)
                got = [tuple(int(v) for v in r) for r in B.hamming_search_dev(needle, (view, hl), n // 3)]
                assert got == want, (shift, n, hl)
    # needle as long as the haystack, and matches at offset 0 and at the last offset
    hay = b"abcdefghij"
    assert [tuple(int(v) for v in r) for r in B.hamming_search_dev(hay, B.haystack_tensor(hay), 0)] == [(0, 10, 0)]
    assert [tuple(int(v) for v in r) for r in B.hamming_search_dev(b"ab", B.haystack_tensor(b"abxxab"), 0)] == [(0, 2, 0), (4, 6, 0)]


def test_filter_path_equals_exact_kernel_on_a_large_shard(monkeypatch):
    """Unit-cost searches with a short needle go through the bit-parallel candidate filter + the exact kernel on the
    flagged blocks.  On a 64 MiB shard the hits must equal the exact kernel's over everything (TA_SEARCH_NOFILTER=1),
    for both cost families, odd needle lengths, k = 0 .. n-1, odd shard lengths and a shifted base."""
    import torch
    from triple_accel_amd import batch as B
    g = Dg.rng(0xF117)
    hay_np = Dg.random_bytes(g, (64 << 20) + 37)
    needles = {n: Dg.random_bytes(g, n).tobytes() for n in (5, 17, 32)}
    for n, needle in needles.items():
        for pos in range(5000 + 100 * n, hay_np.size - 100, 1 << 19):        # a mutated copy every 512 KiB
            mm = np.frombuffer(Dg.mutate(g, needle, max(1, n // 3), True), dtype=np.uint8)
            hay_np[pos:pos + mm.size] = mm
    hay = B.haystack_tensor(hay_np)
    for n, needle in needles.items():
        for k, costs in [(n // 2, (1, 1, 0, None)), (n // 3, (1, 1, 0, 1)), (0, (1, 1, 0, None)), (n - 1, (1, 1, 0, 1))]:
            if k == n - 1 and n > 5:
                continue                                                       # dense: covered below on a small shard
            monkeypatch.delenv("TA_SEARCH_NOFILTER", raising=False)
            got = B.levenshtein_search_dev(needle, hay, k, costs, base=1000, emit_from=1000 + 77)
            monkeypatch.setenv("TA_SEARCH_NOFILTER", "1")
            want = B.levenshtein_search_dev(needle, hay, k, costs, base=1000, emit_from=1000 + 77)
            assert np.array_equal(got, want), (n, k, costs, len(got), len(want))
            assert len(want) > 0 or k < n // 2                                  # the planted copies carry up to n/3 edits
    torch.cuda.synchronize()


def test_filter_falls_back_when_matches_are_dense():
    """k close to the needle length: nearly every block is flagged, the host takes the exact kernel over everything."""
    g = Dg.rng(77)
    needle = g.integers(97, 100, size=6, dtype=np.uint8).tobytes()
    hay = g.integers(97, 100, size=50000, dtype=np.uint8).tobytes()
    for k in (2, 4, 5):
        for costs in [(1, 1, 0, None), (1, 1, 0, 1)]:
            for st in (O.ALL, O.BEST):
                want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, False)
                assert prod_search(needle, hay, k, st, costs) == want, (k, costs, st)


@pytest.mark.parametrize("costs", [(1, 1, 0, None), (1, 1, 0, 1)])
def test_long_needles_through_the_filter(costs, monkeypatch):
    """Needles of 33..512 bytes: multi-dword candidate filter + the memory-backed exact kernel on the flagged blocks,
    against the oracle (All and Best) and against the exact kernel over everything (beyond 256 bytes: 12- / 16-dword vectors, round 5)."""
    from triple_accel_amd import batch as B
    g = Dg.rng(0x10F6)
    for n in (33, 64, 100, 256, 257, 300, 420, 512):
        needle = Dg.rand_str(g, n)
        hay = Dg.planted_haystack(int(g.integers(1 << 30)), needle, 150_000, 9000 + n, max(1, n // 6))
        for k in (n // 8, n // 3):
            for st in (O.ALL, O.BEST):
                want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, False)
                assert prod_search(needle, hay, k, st, costs) == want, (n, k, st, costs)
                assert len(want) > 0
        ht = B.haystack_tensor(hay)
        got = B.levenshtein_search_dev(needle, ht, n // 3, costs)
        monkeypatch.setenv("TA_SEARCH_NOFILTER", "1")
        assert np.array_equal(got, B.levenshtein_search_dev(needle, ht, n // 3, costs))
        monkeypatch.delenv("TA_SEARCH_NOFILTER")


def test_full_size_cfg5_filter_equals_exact(monkeypatch):
    """BASELINE cfg5 at full size (32-byte needle, k = 16, one 1 GiB shard): the filter path returns exactly the hits of
    the exact kernel over the whole shard; every planted copy is found; the first 2 MiB agree with the oracle."""
    import torch
    from triple_accel_amd import batch as B
    g = Dg.rng(0x7C05)
    needle = Dg.random_bytes(Dg.rng(0x7A05), 32).tobytes()
    hay_np = Dg.random_bytes(g, 1 << 30)
    planted = list(range(1 << 16, hay_np.size - 100, 1 << 20))
    for pos in planted:
        mm = np.frombuffer(Dg.mutate(g, needle, 10), dtype=np.uint8)
        hay_np[pos:pos + mm.size] = mm
    hay = B.haystack_tensor(hay_np)
    got = B.levenshtein_search_dev(needle, hay, 16)
    monkeypatch.setenv("TA_SEARCH_NOFILTER", "1")
    want = B.levenshtein_search_dev(needle, hay, 16)
    assert np.array_equal(got, want) and len(got) >= len(planted)
    ends = got[:, 1]
    for pos in planted[::37]:
        assert ((ends > pos) & (ends <= pos + 48)).any(), pos
    ns = 2 << 20
    ora = O.levenshtein_search_naive_with_opts(needle, hay_np[:ns].tobytes(), 16, O.ALL, (1, 1, 0, None), False)
    assert [tuple(int(v) for v in r) for r in got if r[1] <= ns] == [w for w in ora if w[1] > 0]
    torch.cuda.synchronize()


def test_best_hits_selected_on_the_device():
    """ta_search_best_hits_dev: the best-k hits picked on the device + the Best fold == the Best fold over all hits ==
    the oracle's Best, incl. runs of overlapping best hits and a haystack without any hit."""
    from triple_accel_amd import batch as B, dist as TD
    g = Dg.rng(0xBE57)
    for trial in range(12):
        n = int(g.integers(4, 40))
        needle = Dg.rand_str(g, n)
        k = int(g.integers(0, max(1, n // 2) + 1))
        costs = (1, 1, 0, None) if trial % 3 else (1, 1, 0, 1)
        hay = Dg.planted_haystack(int(g.integers(1 << 30)), needle, 200_000, 40, max(1, k))
        if trial == 5:
            hay = (needle * 3 + b"zz" + needle + needle[: n // 2] + needle) * 50            # runs of overlapping exact hits
        if trial == 7:
            hay, k = bytes(Dg.random_bytes(g, 50_000)), 0                                    # (almost surely) no hit at all
        t = B.haystack_tensor(hay)
        allhits = B.levenshtein_search_dev(needle, t, k, costs)
        best_rows = B.levenshtein_search_best_dev(needle, t, k, costs)
        if len(allhits):
            kmin = allhits[:, 2].min()
            assert (best_rows[:, 2] == kmin).all() and len(best_rows) == int((allhits[:, 2] == kmin).sum())
            assert (np.diff(best_rows[:, 1]) > 0).all()
        else:
            assert len(best_rows) == 0
        whole = len(needle) * costs[1] + costs[2]
        head = [(0, 0, whole)] if whole <= k else []
        got = TD.fold_best(head + [tuple(int(v) for v in r) for r in best_rows], k, True)
        assert got == TD.fold_best(head + [tuple(int(v) for v in r) for r in allhits], k, True)
        assert got == O.levenshtein_search_naive_with_opts(needle, hay, k, O.BEST, costs, False), (trial, n, k)


def test_hamming_search_naive_contract():
    """hamming_search_naive_with_opts (src/hamming.rs:96-146): NUL bytes in the haystack are legal, an empty needle matches
    with k = 0 at every offset -- unlike the SIMD-contract entry, which panics / returns nothing there."""
    import triple_accel_amd as T
    g = Dg.rng(19)
    hay = bytearray(g.integers(0, 4, size=4000, dtype=np.uint8).tobytes())      # plenty of NUL bytes
    for n in (1, 2, 5, 33, 40):
        needle = bytes(hay[700:700 + n])
        for k in (0, 1, n // 2):
            for st in (O.ALL, O.BEST):
                want = O.hamming_search_naive_with_opts(needle, bytes(hay), k, st)
                assert [tuple(m) for m in T.hamming_search_naive_with_opts(needle, bytes(hay), k, st)] == want, (n, k, st)
    assert [tuple(m) for m in T.hamming_search_naive_with_opts(b"", b"abc", 2, O.ALL)] == O.hamming_search_naive_with_opts(b"", b"abc", 2, O.ALL)
    with pytest.raises(T.PanicError):       # Best: the reference divides by the needle length (src/hamming.rs:136)
        list(T.hamming_search_naive_with_opts(b"", b"abc", 2, O.BEST))
    assert [tuple(m) for m in T.hamming_search_naive(b"abc", b"  abd")] == [(2, 5, 1)]      # the doc-test of src/hamming.rs:66
    assert list(T.hamming_search_naive(b"abcd", b"abc")) == []


def test_scalar_names_are_the_same_engine():
    """The reference's scalar entry points under their own names: generic items (chars, ints) ride a code table."""
    import triple_accel_amd as T
    assert T.levenshtein_naive(b"abc", b"ab") == 1 and T.levenstein_naive_str("abc", "ab") == 1
    assert T.levenstein_naive_str("naïve café", "naive cafe") == 2
    assert T.levenshtein_naive([10, 2000, 30, 70000], [10, 30, 70000, 5]) == 2
    d, tr = T.levenshtein_naive_with_opts(b"abc", b"ab", True, T.LEVENSHTEIN_COSTS)
    assert d == 1 and [tuple(e) for e in tr] == [("Match", 2), ("BGap", 1)]               # doc-test of src/levenshtein.rs:143
    assert T.levenshtein_naive_k(b"abc", b"ab", 1) == 1 and T.levenshtein_naive_k(b"abc", b"", 1) is None
    assert T.levenshtein_naive_k_with_opts("abcd", "abdc", 1, False, T.RDAMERAU_COSTS) == (1, None)
    assert [tuple(m) for m in T.levenshtein_search_naive(b"abc", b"  abd")] == [tuple(m) for m in T.levenshtein_search(b"abc", b"  abd")]
    with pytest.raises(NotImplementedError):
        T.levenshtein_naive(list(range(300)), list(range(300)))
    with pytest.raises(TypeError):
        T.levenshtein(5, b"abc")
    with pytest.raises(OverflowError):
        T.levenshtein_simd_k(b"a", b"b", 1 << 32)


@pytest.mark.parametrize("costs", [(1, 1, 0, None), (1, 1, 0, 1)])
def test_wavefront_per_block_kernel_lengths_and_edges(costs):
    """The exact kernel behind the filter is one wavefront per flagged block (lev_search_wave_body.h, needles <= 64 bytes):
    every needle length class, matches touching the first and the last byte of the haystack, a partial last block, k from 0
    to n - 1, All and Best, against the oracle."""
    g = Dg.rng(0x3A7E)
    for n in (1, 2, 7, 31, 32, 33, 48, 63, 64):
        needle = Dg.rand_str(g, n)
        for k in sorted({0, n // 3, n // 2, n - 1}):
            body = bytearray(Dg.planted_haystack(int(g.integers(1 << 30)), needle, 9000 + 13 * n, 700 + n, max(1, min(k, n // 2))))
            hay = bytes(needle[1:] + body + needle[:-1] if n > 1 else needle + body + needle)
            if k == n - 1 and n > 7:
                hay = hay[:5000]                                      # nearly everything matches: keep the oracle quick
            for st in (O.ALL, O.BEST):
                want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, False)
                assert prod_search(needle, hay, k, st, costs) == want, (n, k, st, costs)


def test_best_pass_with_more_best_hits_than_the_report_holds():
    """The fused Best pass returns up to 680 best hits in its report; beyond that the selection falls back to
    ta_search_best_hits_dev's path.  1,500 exact copies: the result is still the oracle's."""
    from triple_accel_amd import batch as B, dist as TD
    g = Dg.rng(0xB16)
    needle = Dg.rand_str(g, 20)
    hay = b"".join(needle + Dg.rand_str(g, 45) for _ in range(1500))
    t = B.haystack_tensor(hay)
    rows = B.levenshtein_search_best_dev(needle, t, 6)
    assert len(rows) == 1500 and (rows[:, 2] == 0).all() and (np.diff(rows[:, 1]) > 0).all()
    got = TD.fold_best([tuple(int(v) for v in r) for r in rows], 6, True)
    assert got == O.levenshtein_search_naive_with_opts(needle, hay, 6, O.BEST, (1, 1, 0, None), False)


def test_search_first_is_the_head_of_the_all_mode_list():
    """ta_levenshtein_search_first = `.next()` on the reference's lazy All-mode iterator (src/levenshtein.rs:2282-2420,
    tests/basic_tests.rs:628-632): the first element of the full list -- whatever window it falls into -- or None."""
    import triple_accel_amd as T
    g = Dg.rng(4242)
    assert T.levenshtein_search_first(b"tst", b"testing 123 tasting!", 1) == T.Match(0, 4, 1)          # the KAT's haystack
    assert T.levenshtein_search_first(b"abc", b"", 5) == T.Match(0, 0, 3)                              # end == 0 match first
    assert T.levenshtein_search_first(b"abc", b"xyzxyz", 0) is None
    assert T.levenshtein_search_first(b"", b"abc", 1) is None and T.levenshtein_search_first(b"", b"abc", 1, anchored=True) == T.Match(0, 0, 0)
    for trial in range(12):
        n = int(g.integers(3, 40))
        needle = Dg.rand_str(g, n)
        k = int(g.integers(0, max(1, n // 3) + 1))
        size = int(g.choice([300, 70_000, 400_000, 3_000_000]))
        hay = bytearray(g.integers(33, 127, size=size, dtype=np.uint8).tobytes())
        where = [None, 10, 65_530, 66_000, 330_000, size - n - 3][trial % 6]                           # window edges of 64 KiB / 256 KiB
        if where is not None and where + n + 5 < size:
            hay[where:where + n] = Dg.mutate(g, needle, k)[:n].ljust(n, b"x")
        hay = bytes(hay)
        costs = [T.LEVENSHTEIN_COSTS, T.RDAMERAU_COSTS, T.EditCosts(2, 1, 1, None)][trial % 3]
        want = O.levenshtein_search_naive_with_opts(needle, hay, k, O.ALL, (costs.mismatch_cost, costs.gap_cost, costs.start_gap_cost, costs.transpose_cost), False)
        got = T.levenshtein_search_first(needle, hay, k, costs)
        assert (tuple(got) if got else None) == (want[0] if want else None), (trial, n, k, size, where)
        # the lazy All-mode iterator hands out the same sequence as the eager list
        it = T.levenshtein_search_simd_with_opts(needle, hay, k, T.SearchType.All, costs, False)
        assert [tuple(m) for m in it] == want
    # a dense result (k >= needle_len: every position matches) still has a first element
    hay = bytes(g.integers(33, 127, size=2_000_000, dtype=np.uint8).tobytes())
    assert tuple(T.levenshtein_search_first(b"abcd", hay, 3)) == O.levenshtein_search_naive_with_opts(b"abcd", hay[:5000], 3, O.ALL, (1, 1, 0, None), False)[0]


def test_search_first_on_a_resident_shard():
    """ta_levenshtein_search_first_dev: the first hit of a shard in HBM (positions + base), whatever window it falls into."""
    from triple_accel_amd import batch as B
    g = Dg.rng(515)
    needle = Dg.rand_str(g, 24)
    for where in (None, 5, 65_000, 262_200, 1_400_000, 5_999_900):
        hay = bytearray(g.integers(33, 127, size=6_000_000, dtype=np.uint8).tobytes())
        if where is not None:
            hay[where:where + 24] = Dg.mutate(g, needle, 3)[:24].ljust(24, b"y")
        hay = bytes(hay)
        want = O.levenshtein_search_naive_with_opts(needle, hay, 5, O.ALL, (1, 1, 0, None), False)
        got = B.levenshtein_search_first_dev(needle, B.haystack_tensor(hay), 5, base=1000)
        assert got == ((want[0][0] + 1000, want[0][1] + 1000, want[0][2]) if want else None), where


def test_hamming_search_forms_and_fused_nul_scan(monkeypatch):
    """The three hamming_search kernels behind one entry -- SWAR with 16 offsets per lane (needles <= 64 bytes), bit-sliced counters
    (9..32 bytes, small k), the round-1 forms behind their switches -- against the oracle on the same haystack; the kernel each
    (n, k) takes; and the NUL-byte scan that rides inside the search kernels: a zero byte anywhere -- first byte, last byte, behind the
    last offset, at a tile boundary -- is the SIMD contract's panic (src/hamming.rs:463), for both kernels."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    g = Dg.rng(77)
    hay_np = Dg.random_bytes(g, 700_000)
    hay_np[hay_np == 0] = 1
    for n, k, kern in [(4, 1, "swar16"), (8, 2, "swar16"), (12, 3, "phase_kernel<1,2>"), (16, 7, "phase_kernel<1,3>"), (16, 8, "swar16"),
                       (24, 6, "phase_kernel<2,3>"), (32, 8, "phase_kernel<2,4>"), (32, 2, "phase_kernel<4,2>"), (32, 31, "phase_kernel<1,5>"),
                       (32, 32, "swar16"), (40, 9, "phase_kernel<1,4>"), (64, 16, "phase_kernel<1,5>"), (64, 20, "swar16"), (9, 1, "phase_kernel<2,1>"), (16, 4, "phase_kernel<2,3>"),
                       (70, 10, "phase_kernel<1,4>"), (300, 40, "hamming_search_kernel")]:
        needle = bytes(int(c) or 1 for c in Dg.random_bytes(g, n))
        hay = hay_np.copy()
        nd = np.frombuffer(needle, dtype=np.uint8)
        for pos in list(range(17, hay.size - 2 * n, 9973)) + [0, hay.size - n]:
            hay[pos:pos + n] = nd
            for q in g.integers(0, n, size=int(g.integers(0, k + 2))):
                hay[pos + int(q)] = 7
        dev = B.haystack_tensor(hay)
        want = O.hamming_search_naive_with_opts(needle, hay.tobytes(), k, O.ALL)
        assert len(want) >= 25
        got = [tuple(int(v) for v in r) for r in B.hamming_search_dev(needle, dev, k)]
        assert kern in T.last_kernel_name(), (n, k, T.last_kernel_name())
        assert got == want, (n, k)
        for sw in ("TA_HAMMING_SEARCH_NO_PHASE", "TA_HAMMING_SEARCH_NO_BITS", "TA_HAMMING_SEARCH_SA", "TA_HAMMING_SEARCH_SWAR"):
            if sw == "TA_HAMMING_SEARCH_SA" and n > 32:
                continue
            monkeypatch.setenv(sw, "1")
            assert [tuple(int(v) for v in r) for r in B.hamming_search_dev(needle, dev, k)] == want, (n, k, sw)
            assert "phase" not in T.last_kernel_name()
            if sw == "TA_HAMMING_SEARCH_NO_PHASE" and 9 <= n <= 32 and k < 16:
                assert "bits_kernel" in T.last_kernel_name() or "swar16" in T.last_kernel_name()
            monkeypatch.delenv(sw)
        if n in (8, 24, 40) or (n, k) == (32, 2):
            for zpos in (0, hay.size - 1, hay.size - n + 1, 128 * 5, 262143, 300_001):
                hz = hay.copy()
                hz[zpos] = 0
                with pytest.raises(T.PanicError):
                    B.hamming_search_dev(needle, B.haystack_tensor(hz), k)


def test_hamming_search_phase_form_every_length(monkeypatch):
    """Bit-sliced counters over a subset of the needle's positions (ham_phase_body.h) on the device: every needle length 9..72 plus long
    needles, k from 0 to n / 2, the plan's phase count and every smaller one forced, a haystack whose length is no multiple of four, planted
    near-copies (also across tile boundaries and at both ends); four-letter text, where the filter passes thousands of candidates."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    g = Dg.rng(91)
    base = Dg.random_bytes(g, 300_003)
    base[base == 0] = 1
    dna = (g.integers(0, 4, size=200_001).astype(np.uint8) + 97)
    used = set()
    for n in list(range(9, 73)) + [100, 128, 255, 1000]:
        needle_r = bytes(int(c) or 1 for c in Dg.random_bytes(g, n))
        needle_d = bytes(dna[1000:1000 + n])
        for k in sorted({0, 1, 2, n // 8, n // 4, n // 3, n // 2}):
            for text, needle in ((base, needle_r), (dna, needle_d)):
                if text is dna and (n % 7 or k > 12):
                    continue
                hay = text.copy()
                nd = np.frombuffer(needle, dtype=np.uint8)
                for pos in list(range(31, hay.size - 2 * n, 20011)) + [0, hay.size - n, 65536 - n // 2]:
                    hay[pos:pos + n] = nd
                    for q in g.integers(0, n, size=int(g.integers(0, k + 2))):
                        hay[pos + int(q)] = 7 if text is base else 97
                dev = B.haystack_tensor(hay)
                want = O.hamming_search_naive_with_opts(needle, hay.tobytes(), k, O.ALL)
                for qf in (None, "2", "1"):
                    if qf:
                        monkeypatch.setenv("TA_HAMMING_PHASE_Q", qf)
                    got = [tuple(int(v) for v in r) for r in B.hamming_search_dev(needle, dev, k)]
                    name = T.last_kernel_name()
                    if qf:
                        monkeypatch.delenv("TA_HAMMING_PHASE_Q")
                    assert got == want, (n, k, qf, name, text is dna)
                    if "phase" in name:
                        used.add(name)
                    else:
                        break
    assert {"hamming_search_phase_kernel<%d,%d>" % (q, b) for q, b in ((1, 5), (1, 1), (2, 4), (2, 2), (4, 1), (4, 2))} <= used, used


WEIGHTED = [(2, 2, 0, None), (2, 3, 1, None), (2, 2, 1, 3), (3, 1, 0, None), (1, 2, 0, None), (4, 3, 3, 5), (2, 2, 0, 2), (3, 2, 0, 1)]


@pytest.mark.parametrize("costs", WEIGHTED)
def test_weighted_costs_through_the_superset_filter(costs, monkeypatch):
    """General EditCosts: the unit-cost scan with k' = srch_filter_k flags a superset of the blocks that hold a weighted hit and the
    exact kernels (one wavefront per flagged block up to 64-byte needles, the memory-backed column beyond) price them with the real
    costs.  All and Best against the oracle (planted copies with gap runs and swaps); on the resident shard the hits equal the
    exact kernel's over everything (TA_SEARCH_NOWFILTER=1: round 4's rule), with a shifted base and emit_from."""
    from triple_accel_amd import batch as B
    g = Dg.rng(0x5F17 + costs[0] * 7 + costs[1])
    for n in (6, 17, 32, 48, 64, 100):
        needle = Dg.rand_str(g, n)
        hay = bytearray(Dg.planted_haystack(int(g.integers(1 << 30)), needle, 120_000, 6000 + n, max(1, n // 5)))
        for pos in range(3000, len(hay) - 2 * n, 11_000):
            m = Dg.mutate(g, needle, max(1, n // 8), swaps=True)
            hay[pos:pos + len(m)] = m
        hay = bytes(hay)
        ht = B.haystack_tensor(hay)
        for k in sorted({costs[0], n // 2, n, n + n // 2}):
            for st in (O.ALL, O.BEST):
                want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, False)
                assert prod_search(needle, hay, k, st, costs) == want, (n, k, st, costs)
            monkeypatch.delenv("TA_SEARCH_NOWFILTER", raising=False)
            got = B.levenshtein_search_dev(needle, ht, k, costs, base=500, emit_from=500 + 33)
            monkeypatch.setenv("TA_SEARCH_NOWFILTER", "1")
            ref = B.levenshtein_search_dev(needle, ht, k, costs, base=500, emit_from=500 + 33)
            monkeypatch.delenv("TA_SEARCH_NOWFILTER")
            assert np.array_equal(got, ref), (n, k, costs, len(got), len(ref))
        assert len(O.levenshtein_search_naive_with_opts(needle, hay, n, O.ALL, costs, False)) > 0


def test_weighted_filter_ties_small_alphabet_and_best_selection():
    """Binary / ternary alphabets under weighted costs: hits in nearly every block (the dense fall-back), quirk Q2's ties in the match
    starts, the device-side Best selection == the Best fold over all hits == the oracle's Best."""
    from triple_accel_amd import batch as B, dist as TD
    g = Dg.rng(0x71E5)
    for costs in WEIGHTED:
        for trial in range(4):
            n = int(g.integers(5, 30))
            needle = g.integers(97, 100, size=n, dtype=np.uint8).tobytes()
            hay = g.integers(97, 100, size=30_000, dtype=np.uint8).tobytes()
            k = int(g.integers(1, n + 1))
            for st in (O.ALL, O.BEST):
                want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, False)
                assert prod_search(needle, hay, k, st, costs) == want, (needle, k, st, costs)
        needle = Dg.rand_str(g, 24)
        k = 10
        hay = Dg.planted_haystack(int(g.integers(1 << 30)), needle, 300_000, 4000, 4)
        t = B.haystack_tensor(hay)
        allhits = B.levenshtein_search_dev(needle, t, k, costs)
        best_rows = B.levenshtein_search_best_dev(needle, t, k, costs)
        assert len(allhits) > 0
        kmin = allhits[:, 2].min()
        assert (best_rows[:, 2] == kmin).all() and len(best_rows) == int((allhits[:, 2] == kmin).sum())
        got = TD.fold_best([tuple(int(v) for v in r) for r in best_rows], k, True)
        assert got == O.levenshtein_search_naive_with_opts(needle, hay, k, O.BEST, costs, False), costs


def test_cfg5w_geometry_shard():
    """cfg5's geometry under the three weighted cost sets of bench.py --workload cfg5w (32-byte needle, k = 16, a 4 MiB random
    shard with planted copies): All and Best against the oracle."""
    import triple_accel_amd as T
    g = Dg.rng(0x7A55)
    needle = Dg.random_bytes(g, 32).tobytes()
    hay = bytearray(Dg.random_bytes(g, 4 << 20).tobytes())
    for pos in range(10000, len(hay) - 100, 200_000):
        m = Dg.mutate(g, needle, 6, swaps=True)
        hay[pos:pos + len(m)] = m
    hay = bytes(hay)
    for costs in [(2, 2, 0, None), (2, 3, 1, None), (2, 2, 1, 3)]:
        for st in (O.ALL, O.BEST):
            want = O.levenshtein_search_naive_with_opts(needle, hay, 16, st, costs, False)
            got = [tuple(m) for m in T.levenshtein_search_simd_with_opts(needle, hay, 16, st, T.EditCosts(*costs), False)]
            assert got == want and len(want) > 0, (costs, st)


def test_lazy_iterator_resumes_on_the_resident_prefix(capfd, monkeypatch):
    """The second element of the lazy All-mode iterator does not upload the haystack again: ta_levenshtein_search_resume keeps what
    ta_levenshtein_search_first uploaded (the TA_DEBUG line says how many bytes), sends the rest, and returns the eager sequence -- hits in the
    first window, beyond it, in the last bytes; a different haystack, another search in between or an exhausted first call make it the plain call."""
    import triple_accel_amd as T
    monkeypatch.setenv("TA_DEBUG", "1")
    g = Dg.rng(777)
    needle = Dg.rand_str(g, 20)
    size = 3_000_000
    for where in ([40_000, 2_000_000], [700_000, 700_300, 2_999_900], [2_999_970]):
        hay = bytearray(g.integers(33, 127, size=size, dtype=np.uint8).tobytes())
        for w in where:
            hay[w:w + 20] = needle
        hay = bytes(hay)
        want = O.levenshtein_search_naive_with_opts(needle, hay, 2, O.ALL, (1, 1, 0, None), False)
        assert len(want) >= len(where)
        it = T.levenshtein_search_simd_with_opts(needle, hay, 2, T.SearchType.All, T.LEVENSHTEIN_COSTS, False)
        first = next(it)
        capfd.readouterr()
        rest = list(it)
        err = capfd.readouterr().err
        assert [tuple(first)] + [tuple(m) for m in rest] == want
        assert "search resume:" in err, err[-300:]
        resident = int(err.split("search resume:")[1].split()[0])
        assert 20 <= resident <= size and resident >= where[0], (resident, where)
        assert resident < size or where[0] > 2_900_000
    # another call in between: the plain call (nothing claimed resident), same sequence
    hay2 = bytes(g.integers(33, 127, size=size, dtype=np.uint8).tobytes())
    it = T.levenshtein_search_simd_with_opts(needle, hay, 2, T.SearchType.All, T.LEVENSHTEIN_COSTS, False)
    first = next(it)
    assert T.levenshtein_search_first(needle, hay2, 2) is None or True
    capfd.readouterr()
    rest = list(it)
    assert "search resume:" not in capfd.readouterr().err
    assert [tuple(first)] + [tuple(m) for m in rest] == want
