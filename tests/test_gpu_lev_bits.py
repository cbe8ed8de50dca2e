"""-m gpu: the bit-parallel band kernel (lev_bits_body.h: unit-cost families, through the C ABI) against the CPU
oracle bit for bit, and against the DP band kernel at BASELINE sizes (two independent HIP implementations of the
same contract must agree on every pair)."""
import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O
from test_gpu_lev_batch import gpu_k, oracle_k, ragged_pairs

pytestmark = pytest.mark.gpu

LEV, RDAM = (1, 1, 0, None), (1, 1, 0, 1)


def kernel_id():
    import triple_accel_amd as T
    return T.last_launch_info()["kernel"]


@pytest.mark.parametrize("costs", [LEV, RDAM])
def test_bits_ragged_vs_oracle(costs):
    a, b = ragged_pairs(21, 6000, 80, 10, costs[3] is not None)
    for k in (0, 1, 2, 3, 7, 12, 30, 47, 61, 64, 100, 125):
        got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
        assert kernel_id() == 3
        assert np.array_equal(got, want), (k, costs, np.flatnonzero(got != want)[:10])


def test_bits_every_window_width(monkeypatch):
    """Every instantiated window width, in the sliding form (TA_BITS_STATIC=1) and, from 8 dwords on, the static form (=2)."""
    a, b = ragged_pairs(22, 3000, 90, 9, True)
    for na in list(range(1, 17)) + list(range(18, 33, 2)):
        monkeypatch.setenv("TA_FORCE_NA", str(na))
        for mode in ("1", "2") if na >= 8 else ("1",):
            monkeypatch.setenv("TA_BITS_STATIC", mode)
            for costs in (LEV, RDAM):
                k = max(0, min(4 * na - 4 - (2 if costs[3] else 0), 9))
                got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
                assert kernel_id() == 3
                assert np.array_equal(got, want), (na, mode, k, costs, np.flatnonzero(got != want)[:10])


@pytest.mark.parametrize("costs", [LEV, RDAM])
def test_bits_stride8_every_tail_length(costs):
    """Fixed-length batches whose length leaves 0..7 columns for the last block of eight (the partly run block is its own code path:
    round 3's SDWA compare of the 33rd diagonal clobbered the scalar condition code there until its asm statement said so)."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    trans = costs[3] is not None
    ec = T.EditCosts(*costs)
    for L in range(296, 304):
        fa, fb = Dg.pairs_mutated_fixed(0x60 + L, 2500, L, 30, trans)
        for k in (32 - (2 if trans else 0), 27):
            got = B.levenshtein_k_batch(B.Strings.from_fixed(fa), B.Strings.from_fixed(fb), k, ec).cpu().numpy().view(np.uint32)
            assert kernel_id() == 3 and T.last_launch_info()["diags_per_lane"] == 33
            want = O.levenshtein_k_batch(O.csr_from_fixed(fa), O.csr_from_fixed(fb), k, costs)
            assert np.array_equal(got, want), (L, k, costs, np.flatnonzero(got != want)[:10])


@pytest.mark.parametrize("costs", [LEV, RDAM])
def test_bits_stride8_window_form(monkeypatch, costs):
    """The stride-8 form (bands of 25..33 diagonals by the planner's choice, narrower ones when forced) on ragged CSR batches -- the
    chunk form of the fetch, pairs ending inside a block of 8 columns while others run on -- and on fixed-length batches of 300-byte
    strings -- the line form -- against the oracle and against the sliding form."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    trans = costs[3] is not None
    a, b = ragged_pairs(23, 5000, 300, 28, trans)
    fa, fb = Dg.pairs_mutated_fixed(0x58, 3000, 300, 30, trans)
    ec = T.EditCosts(*costs)
    kmax = 32 - (2 if trans else 0)
    for k in (kmax, kmax - 1, 26, 24 - (2 if trans else 0)):              # 33, 32, 27 / 29, 25 diagonals: the planner's own choice
        got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
        assert kernel_id() == 3 and T.last_launch_info()["diags_per_lane"] == 33
        assert np.array_equal(got, want), ("csr", k, costs, np.flatnonzero(got != want)[:10])
        got = B.levenshtein_k_batch(B.Strings.from_fixed(fa), B.Strings.from_fixed(fb), k, ec).cpu().numpy().view(np.uint32)
        assert kernel_id() == 3 and T.last_launch_info()["diags_per_lane"] == 33
        want = O.levenshtein_k_batch(O.csr_from_fixed(fa), O.csr_from_fixed(fb), k, costs)
        assert np.array_equal(got, want), ("fixed", k, costs, np.flatnonzero(got != want)[:10])
        assert (want != 0xFFFFFFFF).mean() > 0.1
    for k in (0, 3, 11, kmax):
        monkeypatch.setenv("TA_BITS_STATIC", "3")
        s8 = gpu_k(a, b, k, costs)
        assert T.last_launch_info()["diags_per_lane"] == 33
        monkeypatch.setenv("TA_BITS_STATIC", "1")
        sl = gpu_k(a, b, k, costs)
        monkeypatch.delenv("TA_BITS_STATIC")
        assert np.array_equal(s8, sl) and np.array_equal(s8, oracle_k(a, b, k, costs)), (k, costs)


def test_bits_static_equals_sliding_on_long_ragged_strings(monkeypatch):
    g = Dg.rng(0x57A7)
    a, b = [], []
    for n in (3, 64, 65, 255, 256, 257, 1000, 3001):
        x = Dg.rand_str(g, n)
        a += [x, x, x[: n // 2]]
        b += [Dg.mutate(g, x, 30, True), Dg.rand_str(g, n + 2), x]
    for k, costs in [(32, LEV), (33, LEV), (35, RDAM), (60, LEV), (100, RDAM), (125, LEV)]:
        monkeypatch.setenv("TA_BITS_STATIC", "2")
        st = gpu_k(a, b, k, costs)
        assert kernel_id() == 3
        monkeypatch.setenv("TA_BITS_STATIC", "1")
        sl = gpu_k(a, b, k, costs)
        assert np.array_equal(st, sl) and np.array_equal(st, oracle_k(a, b, k, costs)), (k, costs)


def test_bits_chunk_lengths_and_long_strings(monkeypatch):
    g = Dg.rng(5)
    a, b = [], []
    for n in (1, 31, 32, 33, 63, 64, 65, 200, 513, 1500, 4000):
        x = Dg.rand_str(g, n)
        a += [x, x, x, x[: n // 2]]
        b += [Dg.mutate(g, x, 25, True), x, Dg.rand_str(g, n + 3), x]
    for ch in (16, 32, 64):
        monkeypatch.setenv("TA_FORCE_CH", str(ch))
        for k, costs in [(30, LEV), (40, RDAM), (61, LEV), (3, RDAM)]:
            got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
            assert kernel_id() == 3
            assert np.array_equal(got, want), (ch, k, costs, np.flatnonzero(got != want)[:10])


def test_unit_costs_beyond_the_window_use_the_dp_kernel():
    """unit_k > 127 (125 with transpositions) does not fit the 128-bit window: the planner falls back to the DP band."""
    a, b = ragged_pairs(23, 2000, 400, 30, True)
    for k, costs in [(128, LEV), (126, RDAM), (250, LEV)]:
        got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
        assert kernel_id() == 1
        assert np.array_equal(got, want), (k, costs)


def test_dp_kernel_still_covers_unit_costs(monkeypatch):
    """TA_NO_BITS=1 routes the unit-cost families through the general DP band kernel (planner's own layout)."""
    monkeypatch.setenv("TA_NO_BITS", "1")
    a, b = ragged_pairs(24, 5000, 80, 10, True)
    for k in (0, 3, 12, 32, 61, 100):
        for costs in (LEV, RDAM):
            got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
            assert kernel_id() == 1
            assert np.array_equal(got, want), (k, costs, np.flatnonzero(got != want)[:10])


@pytest.mark.parametrize("wl", ["cfg2", "cfg4"])
def test_full_size_bits_equals_dp(wl, monkeypatch):
    """BASELINE cfg2 / cfg4 at full size (1M pairs): the bit-parallel kernel and the DP band kernel agree on every
    pair (half random, half mutated so that both None and Some(d) are exercised), plus a sampled oracle check."""
    import torch
    from triple_accel_amd import batch as B
    n, L, k, costs = {"cfg2": (1_000_000, 256, 32, LEV), "cfg4": (1_000_000, 128, 8, RDAM)}[wl]
    ar, br = Dg.pairs_random(0x7B00 + L, n // 2, L)
    am, bm = Dg.pairs_mutated_fixed(0x7B10 + L, n // 2, L, k, swaps=costs[3] is not None)
    a, b = np.concatenate([ar, am]), np.concatenate([br, bm])
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    bits = B.levenshtein_k_batch(sa, sb, k, costs).cpu().numpy().view(np.uint32)
    assert kernel_id() == 3
    monkeypatch.setenv("TA_NO_BITS", "1")
    dp = B.levenshtein_k_batch(sa, sb, k, costs).cpu().numpy().view(np.uint32)
    assert kernel_id() == 1
    assert np.array_equal(bits, dp), np.flatnonzero(bits != dp)[:10]
    assert (bits[: n // 2] == 0xFFFFFFFF).all() and (bits[n // 2:] != 0xFFFFFFFF).mean() > 0.2
    idx = np.concatenate([np.arange(0, 2000), np.arange(n - 4000, n)])
    want = O.levenshtein_k_batch(O.csr_from_fixed(a[idx]), O.csr_from_fixed(b[idx]), k, costs)
    assert np.array_equal(bits[idx], want)
    torch.cuda.synchronize()


def long_pairs(seed, lens, kmut):
    g = Dg.rng(seed)
    a, b = [], []
    for n in lens:
        x = Dg.rand_str(g, n)
        a += [x, x, x, x[: n // 3], x]
        b += [Dg.mutate(g, x, kmut, True), Dg.rand_str(g, max(0, n - 17)), x, x, Dg.mutate(g, x, 3 * kmut, True)]
    return a, b


@pytest.mark.parametrize("rows", [32, 64])
def test_widebits_forced_vs_oracle(rows, monkeypatch):
    """The row-blocked bit-parallel kernel on everything it may be given: short to 2 KiB / 4 KiB strings, both cost
    families, bounded and unbounded k."""
    monkeypatch.setenv("TA_FORCE_WIDEBITS", str(rows))
    top = 2048 if rows == 32 else 4096
    a, b = long_pairs(31 + rows, (1, 5, 31, 32, 33, 64, 65, 200, 777, top // 2 + 1, top - 200), 40)   # mutations may add up to 120 bytes
    a += [Dg.rand_str(Dg.rng(1), top)]; b += [Dg.rand_str(Dg.rng(2), top)]
    a += [b"", b"", b"\0" * 100]; b += [b"", b"abc", b"\0" * 90 + b"\1"]
    for k, costs in [(0xFFFFFFFF, LEV), (0xFFFFFFFF, RDAM), (150, LEV), (130, RDAM), (1000, LEV)]:
        got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
        assert kernel_id() == 4
        assert np.array_equal(got, want), (rows, k, costs, np.flatnonzero(got != want)[:10])


def test_widebits_small_alphabet(monkeypatch):
    monkeypatch.setenv("TA_FORCE_WIDEBITS", "32")
    g = Dg.rng(8)
    a = [bytes(g.integers(97, 100, size=int(g.integers(0, 900))).astype(np.uint8)) for _ in range(300)]
    b = [bytes(g.integers(97, 100, size=int(g.integers(0, 900))).astype(np.uint8)) for _ in range(300)]
    for costs in (LEV, RDAM):
        got, want = gpu_k(a, b, 0xFFFFFFFF, costs), oracle_k(a, b, 0xFFFFFFFF, costs)
        assert kernel_id() == 4
        assert np.array_equal(got, want), (costs, np.flatnonzero(got != want)[:10])


def test_widebits_several_stripes(monkeypatch):
    """Strings longer than one stripe on both sides (boundary lines in HBM, anchor chain, band-limited columns), with the
    stripe height forced to 2048 and 4096 rows; includes the lengths right at the stripe edges."""
    g = Dg.rng(0x57)
    a, b = [], []
    for n in (2047, 2048, 2049, 4095, 4096, 4097, 6000, 9000, 12500):
        x = Dg.rand_str(g, n)
        a += [x, x, x]
        b += [Dg.mutate(g, x, 150, True), Dg.rand_str(g, n - 100), x[: n - 300]]
    a += [bytes(g.integers(97, 100, size=7000).astype(np.uint8))]; b += [bytes(g.integers(97, 100, size=6800).astype(np.uint8))]
    for rows in (32, 64):
        monkeypatch.setenv("TA_FORCE_WIDEBITS", str(rows))
        for k, costs in [(0xFFFFFFFF, LEV), (0xFFFFFFFF, RDAM), (400, LEV), (170, RDAM)]:
            got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
            assert kernel_id() == 4
            assert np.array_equal(got, want), (rows, k, costs, np.flatnonzero(got != want)[:10])


def test_long_unit_cost_pairs_pick_widebits_by_themselves():
    import triple_accel_amd as T
    g = Dg.rng(3)
    x = Dg.rand_str(g, 3000)
    y = Dg.mutate(g, x, 100, True)
    assert T.levenshtein(x, y) == O.levenshtein(x, y)
    assert kernel_id() == 4
    assert T.rdamerau(x, y) == O.rdamerau(x, y)
    assert kernel_id() == 4
    z = Dg.rand_str(g, 2800)
    assert T.levenshtein_exp(x, z) == O.levenshtein_exp(x, z)
    assert T.rdamerau_exp(z, x) == O.rdamerau_exp(z, x)
    x2, y2 = Dg.rand_str(g, 20000), Dg.rand_str(g, 19000)                      # 5 stripes of 4096 rows
    assert T.levenshtein(x2, y2) == O.levenshtein(x2, y2)
    assert kernel_id() == 4


def test_widebits_equals_dp_on_cfg3_shape(monkeypatch):
    """BASELINE cfg3 geometry (4 KiB pairs): levenshtein_exp through the bit-parallel kernels == through the DP kernels
    == oracle on a sample (the DP path is the slow one: keep the batch small)."""
    from triple_accel_amd import batch as B
    n = 192
    ar, br = Dg.pairs_random(0x7B03, n // 2, 4096)
    am, bm = Dg.pairs_mutated_fixed(0x7B13, n // 2, 4096, 300)
    a, b = np.concatenate([ar, am]), np.concatenate([br, bm])
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    bits = B.levenshtein_exp_batch(sa, sb).cpu().numpy().view(np.uint32)
    monkeypatch.setenv("TA_NO_BITS", "1")
    dp = B.levenshtein_exp_batch(sa, sb).cpu().numpy().view(np.uint32)
    assert np.array_equal(bits, dp)
    idx = np.r_[0:8, n - 8:n]
    want = O.levenshtein_exp_batch(O.csr_from_fixed(a[idx]), O.csr_from_fixed(b[idx]))
    assert np.array_equal(bits[idx], want)


def test_full_size_cfg3_bits_equals_dp(monkeypatch):
    """BASELINE cfg3 at full size (100K x 4 KiB, levenshtein_exp): the bit-parallel schedule (band rounds + row-blocked
    kernel) and the DP schedule (band rounds + DP band / wide kernels) return the same distance for every pair; identical
    pairs give 0, and d(a, b) == d(b, a)."""
    import torch
    from triple_accel_amd import batch as B
    n, L = 100_000, 4096
    ar, br = Dg.pairs_random(0x7C03, n // 2, L)
    am, bm = Dg.pairs_mutated_fixed(0x7C13, n // 2, L, 200)
    a, b = np.concatenate([ar, am]), np.concatenate([br, bm])
    b[n - 100:] = a[n - 100:]                                                    # some identical pairs
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    bits = B.levenshtein_exp_batch(sa, sb).cpu().numpy().view(np.uint32)
    swapped = B.levenshtein_exp_batch(sb, sa).cpu().numpy().view(np.uint32)
    assert np.array_equal(bits, swapped)
    assert (bits[n - 100:] == 0).all() and (bits[: n // 2] > 3500).all() and (bits[n // 2: n - 100] <= 400).all()      # <= 200 edits, then cut or padded back to 4096 bytes
    monkeypatch.setenv("TA_NO_BITS", "1")
    dp = B.levenshtein_exp_batch(sa, sb).cpu().numpy().view(np.uint32)
    assert np.array_equal(bits, dp), np.flatnonzero(bits != dp)[:10]
    idx = np.r_[0:4, n // 2: n // 2 + 4]
    assert np.array_equal(bits[idx], O.levenshtein_exp_batch(O.csr_from_fixed(a[idx]), O.csr_from_fixed(b[idx])))
    torch.cuda.synchronize()


def test_unforced_dispatch_sweep():
    """Whatever kernel lev_choose picks (bit-parallel band, row-blocked bit-parallel, DP band, DP wide), the answer is the
    oracle's: string lengths from 60 to 6000, k from tight to unbounded, unit and weighted cost families."""
    g = Dg.rng(0xD15)
    seen = set()
    for n in (60, 130, 260, 600, 1100, 2100, 3000, 6000):
        x = Dg.rand_str(g, n)
        a = [x, x, x, x[: n - 7]]
        b = [Dg.mutate(g, x, max(2, n // 25), True), Dg.rand_str(g, n), x, x]
        for costs in [LEV, RDAM, (2, 1, 0, None), (1, 1, 1, 1)]:
            for k in (n // 20, 127, 128, n // 2, 0xFFFFFFFF):
                got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
                seen.add(kernel_id())
                assert np.array_equal(got, want), (n, k, costs, kernel_id(), got, want)
    assert seen == {1, 2, 3, 4}, seen


def test_one_long_pair_is_spread_over_many_wavefronts(monkeypatch):
    """The single-call API on long strings: the pair's stripes are cut into tiles and launched diagonal by diagonal.  Same
    answers as the one-wavefront form (TA_WB_NO_TILES=1) and as the oracle, for several tile lengths, both cost families,
    bounded k (band-limited stripes) and the exp loop."""
    import triple_accel_amd as T
    g = Dg.rng(0x7117)
    x = Dg.rand_str(g, 30000)
    y = Dg.mutate(g, x, 700, True)
    z = Dg.rand_str(g, 26000)
    for tile in ("", "64", "1024"):
        if tile:
            monkeypatch.setenv("TA_WB_TILE_STEPS", tile)
        for a, b in ((x, y), (y, x), (x, z)):
            assert T.levenshtein(a, b) == O.levenshtein(a, b), (tile, len(a), len(b))
            assert kernel_id() == 4 and T.last_launch_info()["grid"] > 10        # `grid` reports the number of launches here
        assert T.rdamerau(x, y) == O.rdamerau(x, y)
        assert T.levenshtein_simd_k(x, y, 2000) == O.levenshtein_simd_k_with_opts(x, y, 2000, False, LEV)[0]
        assert T.levenshtein_simd_k(x, y, 500) == O.levenshtein_simd_k_with_opts(x, y, 500, False, LEV)[0] == 490   # band barely wide enough
        assert T.levenshtein_simd_k(x, y, 489) is None
    monkeypatch.delenv("TA_WB_TILE_STEPS")
    assert T.levenshtein_exp(x, y) == O.levenshtein_exp(x, y)
    d_tiles = T.levenshtein(x, z)
    monkeypatch.setenv("TA_WB_NO_TILES", "1")
    assert T.levenshtein(x, z) == d_tiles


def test_a_few_long_pairs_in_a_fixed_length_batch():
    """Up to 16 long fixed-length pairs take the tiled form pair after pair; more of them the one-wavefront-per-pair form."""
    from triple_accel_amd import batch as B
    for n in (3, 20):
        am, bm = Dg.pairs_mutated_fixed(0x7C20 + n, n, 10000, 300)
        got = B.levenshtein_k_batch(B.Strings.from_fixed(am), B.Strings.from_fixed(bm), 0xFFFFFFFF).cpu().numpy().view(np.uint32)
        assert kernel_id() == 4
        want = O.levenshtein_k_batch(O.csr_from_fixed(am), O.csr_from_fixed(bm), 0xFFFFFFFF)
        assert np.array_equal(got, want), (n, got, want)


def test_small_passes_are_chosen_by_wavefront_latency(monkeypatch):
    """Up to 1024 pairs the pass lasts as long as its slowest wavefront: 256-byte pairs go through the row-blocked
    kernel (one pair per wavefront) although the band kernel (64 pairs per wavefront) is cheaper per pair, and a LONE pair
    whose band fits 64 diagonals through the single-pair kernel (match vectors 64 columns at a time, the recurrence on the
    scalar unit); the results are the oracle's either way, and TA_NO_LATENCY_RULE=1 restores the throughput choice."""
    import triple_accel_amd as T
    g = Dg.rng(0x1A7)
    for n_pairs in (1, 40, 900):
        a, b = [], []
        for i in range(n_pairs):
            x = Dg.rand_str(g, int(g.integers(200, 300)))
            a.append(x); b.append(Dg.mutate(g, x, int(g.integers(0, 45)), True) if i % 3 else Dg.rand_str(g, len(x)))
        for k, costs in [(32, LEV), (32, RDAM), (8, LEV), (0xFFFFFFFF, LEV), (0xFFFFFFFF, RDAM)]:
            monkeypatch.delenv("TA_NO_LATENCY_RULE", raising=False)
            got = gpu_k(a, b, k, costs)
            kid = kernel_id()
            want = oracle_k(a, b, k, costs)
            assert np.array_equal(got, want), (n_pairs, k, costs)
            if n_pairs == 1 and k <= 32:
                assert kid == 6, (k, costs, kid)          # a lone pair, band <= 64 diagonals: the single-pair kernel (lev_one_body.h)
            else:
                assert (kid in (3, 4)) if k == 8 else kid == 4, (n_pairs, k, costs, kid)
            monkeypatch.setenv("TA_NO_LATENCY_RULE", "1")
            assert np.array_equal(gpu_k(a, b, k, costs), want)
            assert kernel_id() == (3 if k <= 32 else 1)
    x = Dg.rand_str(g, 256)
    y = Dg.mutate(g, x, 10, True)
    monkeypatch.delenv("TA_NO_LATENCY_RULE", raising=False)
    assert T.levenshtein(x, y) == O.levenshtein(x, y) and kernel_id() == 4
    assert T.rdamerau(x, y) == O.rdamerau(x, y) and kernel_id() == 4


def test_byte_values_around_the_perm_selector_codes():
    """The match vector tests bytes with v_perm_b32 selectors (a ^ b ^ 0x0C == 12): alphabets made of the selector codes that
    behave specially (0..13, the sign-replicating 8..11, 0x0C itself, 0xFF) and the full byte range must still compare
    exactly, in the sliding and the static window forms, with and without transpositions."""
    g = Dg.rng(0xC0DE)
    alphabets = [[0x0C, 0x00], [0x0C, 0x0D], [0x00, 0x08, 0x0B, 0x0C], [0x0C, 0x8C, 0xFF, 0x04], list(range(0, 16)), list(range(256))]
    for alpha in alphabets:
        al = np.array(alpha, dtype=np.uint8)
        a, b = [], []
        for _ in range(1500):
            n = int(g.integers(0, 120))
            x = al[g.integers(0, len(al), n)]
            if g.random() < 0.7:
                y = x.copy()
                for _e in range(int(g.integers(0, 10))):
                    if len(y) == 0:
                        break
                    op, pos = int(g.integers(0, 4)), int(g.integers(0, len(y)))
                    if op == 0:
                        y[pos] = al[int(g.integers(0, len(al)))]
                    elif op == 1:
                        y = np.insert(y, pos, al[int(g.integers(0, len(al)))])
                    elif op == 2:
                        y = np.delete(y, pos)
                    elif pos + 1 < len(y):
                        y[pos], y[pos + 1] = y[pos + 1], y[pos]
            else:
                y = al[g.integers(0, len(al), int(g.integers(0, 120)))]
            a.append(x.tobytes()); b.append(y.tobytes())
        for k, costs in [(3, LEV), (9, RDAM), (30, LEV), (33, RDAM), (60, LEV)]:
            got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
            assert kernel_id() == 3
            assert np.array_equal(got, want), (alpha[:4], k, costs, np.flatnonzero(got != want)[:10])


@pytest.mark.parametrize("L,Lb,k,costs", [(128, 128, 8, (1, 1, 0, 1)), (128, 128, 8, (1, 1, 0, None)), (96, 100, 12, (1, 1, 0, None)),
                                          (64, 61, 10, (1, 1, 0, 1)), (200, 200, 14, (1, 1, 0, None))])
def test_two_pairs_per_lane_narrow_bands(monkeypatch, L, Lb, k, costs):
    """Big fixed-length batches with a band of at most 15 diagonals run two pairs per lane (lev_bits2_body.h, 128 pairs per
    wavefront): vs the oracle on a sample, and pair by pair against the one-pair-per-lane kernel (TA_NO_BITS2=1)."""
    import torch
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    n = 300_000 + 77                                   # a ragged last wavefront
    g = Dg.rng(L * 3 + k)
    a = g.integers(97, 101, size=(n, L), dtype=np.uint8)
    b = g.integers(97, 101, size=(n, Lb), dtype=np.uint8)
    m = min(L, Lb)
    sim = g.random(n) < 0.8
    b[sim, :m] = a[sim, :m]
    for row in np.nonzero(sim)[0][:20000]:
        s = Dg.mutate(g, bytes(b[row]), int(g.integers(0, k + 3)), swaps=costs[3] is not None)
        s = (s + bytes(g.integers(97, 101, size=Lb, dtype=np.uint8)))[:Lb]
        b[row] = np.frombuffer(s, dtype=np.uint8)
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    got = B.levenshtein_k_batch(sa, sb, k, costs).cpu().numpy().view(np.uint32)
    info = T.last_launch_info()
    assert info["kernel"] == 3 and info["pairs_per_wave"] == 128, info
    ns = 30000
    want = O.levenshtein_k_batch(O.csr_from_fixed(a[:ns]), O.csr_from_fixed(b[:ns]), k, costs)
    assert np.array_equal(got[:ns], want)
    tail = O.levenshtein_k_batch(O.csr_from_fixed(a[-500:]), O.csr_from_fixed(b[-500:]), k, costs)
    assert np.array_equal(got[-500:], tail)
    assert (want != 0xFFFFFFFF).any() and (want == 0xFFFFFFFF).any()
    monkeypatch.setenv("TA_NO_BITS2", "1")
    one = B.levenshtein_k_batch(sa, sb, k, costs).cpu().numpy().view(np.uint32)
    assert T.last_launch_info()["pairs_per_wave"] == 64
    assert np.array_equal(got, one)


def test_two_pairs_per_lane_is_for_big_batches_only(monkeypatch):
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    a, b = Dg.pairs_mutated_fixed(5, 5000, 128, 8)
    B.levenshtein_k_batch(B.Strings.from_fixed(a), B.Strings.from_fixed(b), 8)
    assert T.last_launch_info()["pairs_per_wave"] == 64


@pytest.mark.parametrize("costs", [LEV, RDAM])
def test_single_pair_kernel(monkeypatch, costs):
    """Single calls with a band of up to 64 diagonals: lev_one_body.h against the oracle and against the kernels it replaces
    (TA_NO_ONE=1), over lengths around the 64-column blocks, all window widths, empty strings, NUL bytes."""
    import triple_accel_amd as T
    g = Dg.rng(0x0E1 + (costs[3] or 0))
    C = T.EditCosts(*costs)
    seen = 0
    for rnd in range(250):
        la = int(g.choice([0, 1, 3, 16, 63, 64, 65, 127, 128, 129, 256, 1000, 4096]))
        a = g.integers(0, 4, size=la, dtype=np.uint8).tobytes() if rnd % 5 == 0 else Dg.rand_str(g, la)
        b = Dg.mutate(g, a, int(g.integers(0, 50)), True) if g.random() < 0.7 else Dg.rand_str(g, int(g.integers(0, la + 30)))
        if not a and not b:
            continue
        k = int(g.choice([0, 1, 5, 8, 16, 31, 32, 33, 48, 60, 61, 62, 63]))
        monkeypatch.delenv("TA_NO_ONE", raising=False)
        got = T.levenshtein_simd_k_with_opts(a, b, k, False, C)
        kid = kernel_id()
        want = O.levenshtein_naive_k_with_opts(a, b, k, False, costs)[0]
        assert (None if got is None else got[0]) == want, (a, b, k, costs)
        if kid == 6:
            seen += 1
            monkeypatch.setenv("TA_NO_ONE", "1")
            other = T.levenshtein_simd_k_with_opts(a, b, k, False, C)
            assert kernel_id() != 6 and (None if other is None else other[0]) == want
    assert seen > 150
    monkeypatch.delenv("TA_NO_ONE", raising=False)
    assert T.levenshtein(b"kitten", b"sitting") == 3 and kernel_id() == 6          # unbounded k on short strings clamps into the band


@pytest.mark.parametrize("L,k,costs", [(256, 32, LEV), (128, 8, RDAM), (200, 20, RDAM)])
def test_early_out_option_same_answers(L, k, costs):
    """ta_set_option(TA_OPT_EARLY_OUT): wavefronts of far pairs stop after a few dozen columns, wavefronts that hold a near pair
    run on -- the answers are those of the default pass, pair by pair, and the oracle's on a sample."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    n = 300_000 + 33
    g = Dg.rng(L + k)
    a = g.integers(33, 127, size=(n, L), dtype=np.uint8)
    b = g.integers(33, 127, size=(n, L), dtype=np.uint8)
    near = g.random(n) < 0.02                                                  # most wavefronts hold far pairs only
    near[:20000] = g.random(20000) < 0.5
    b[near] = a[near]
    pos = g.integers(0, L, size=(n, max(1, k // 2)))
    rows = np.nonzero(near)[0]
    b[rows[:, None], pos[rows]] = 32
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    base = B.levenshtein_k_batch(sa, sb, k, costs).cpu().numpy().view(np.uint32)
    try:
        T.set_option(T.OPT_EARLY_OUT, True)
        got = B.levenshtein_k_batch(sa, sb, k, costs).cpu().numpy().view(np.uint32)
    finally:
        T.set_option(T.OPT_EARLY_OUT, False)
    assert np.array_equal(got, base)
    ns = 20000
    assert np.array_equal(got[:ns], O.levenshtein_k_batch(O.csr_from_fixed(a[:ns]), O.csr_from_fixed(b[:ns]), k, costs))
    assert (base != 0xFFFFFFFF).sum() > 1000 and (base == 0xFFFFFFFF).sum() > 1000


@pytest.mark.parametrize("L,Lb,k,costs,alphabet", [(256, 256, 32, LEV, b"ACGT"), (128, 128, 8, RDAM, b"ACGT"), (200, 190, 30, RDAM, b"acgu"),
                                                   (96, 100, 20, LEV, b"TG")])
def test_small_alphabet_kernel(L, Lb, k, costs, alphabet):
    """levenshtein_k_batch(..., alphabet=...): the table-lookup kernel (lev_bitsq_body.h) against the oracle on a sample and pair by
    pair against the byte-test kernels; pairs that hold a byte outside the alphabet (here: 'N', first / last / middle positions)
    are answered by the general kernel inside the same call."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    n = 200_000 + 21
    g = Dg.rng(L + Lb + k)
    sym = np.frombuffer(alphabet, dtype=np.uint8)
    a = sym[g.integers(0, len(sym), size=(n, L))]
    b = sym[g.integers(0, len(sym), size=(n, Lb))]
    m = min(L, Lb)
    near = g.random(n) < 0.6
    b[near, :m] = a[near, :m]
    for row in np.nonzero(near)[0][:20000]:
        s = Dg.mutate(g, bytes(b[row]), int(g.integers(0, k + 3)), swaps=costs[3] is not None)
        s = bytes(int(sym[c % len(sym)]) if c not in sym else c for c in s)       # mutate() writes spaces / printable bytes: fold them into the alphabet
        s = (s + sym[g.integers(0, len(sym), size=Lb)].tobytes())[:Lb]
        b[row] = np.frombuffer(s, dtype=np.uint8)
    foreign = g.choice(n, size=300, replace=False)
    for t, row in enumerate(foreign):
        (a if t % 2 else b)[row, [0, (Lb if t % 2 == 0 else L) - 1, m // 2][t % 3]] = ord("N")
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    got = B.levenshtein_k_batch(sa, sb, k, costs, alphabet=alphabet).cpu().numpy().view(np.uint32)
    info = T.last_launch_info()
    assert info["kernel"] == 7 and T.last_kernel_name().startswith("lev_bitsq_kernel<"), (info, T.last_kernel_name())
    base = B.levenshtein_k_batch(sa, sb, k, costs).cpu().numpy().view(np.uint32)
    assert T.last_launch_info()["kernel"] == 3
    assert np.array_equal(got, base), np.flatnonzero(got != base)[:10]
    ns = 20000
    assert np.array_equal(got[:ns], O.levenshtein_k_batch(O.csr_from_fixed(a[:ns]), O.csr_from_fixed(b[:ns]), k, costs))
    fr = np.sort(foreign)[:200]
    assert np.array_equal(got[fr], O.levenshtein_k_batch(O.csr_from_fixed(a[fr]), O.csr_from_fixed(b[fr]), k, costs))
    assert (base != 0xFFFFFFFF).sum() > 1000 and (base == 0xFFFFFFFF).sum() > 1000
    # what the kernels do not cover runs the general path: wide bands, general costs, alphabets without a code (two cases of the letters;
    # more than 32 symbols)
    for kk, cc, al in [(40, costs, alphabet), (k, (2, 3, 1, None), alphabet), (k, costs, b"ACGTacgt"), (k, costs, bytes(range(40)))]:
        out = B.levenshtein_k_batch(sa, sb, kk, cc, alphabet=al).cpu().numpy().view(np.uint32)
        assert T.last_launch_info()["kernel"] != 7
        assert np.array_equal(out[:2000], O.levenshtein_k_batch(O.csr_from_fixed(a[:2000]), O.csr_from_fixed(b[:2000]), kk, cc))


PROTEIN = b"ACDEFGHIKLMNPQRSTVWY"
IUPAC = b"ACGTRYSWKMBDHVNU"


@pytest.mark.parametrize("L,Lb,k,costs,alphabet", [(256, 256, 32, LEV, PROTEIN), (128, 128, 8, RDAM, IUPAC), (200, 190, 30, RDAM, PROTEIN),
                                                   (96, 100, 20, LEV, b"0123456789"), (256, 256, 32, LEV, b"ACGTN"), (300, 310, 25, RDAM, bytes(range(0x60, 0x80)))])
def test_alphabets_of_up_to_32_symbols(L, Lb, k, costs, alphabet, monkeypatch):
    """levenshtein_k_batch(..., alphabet=...) with 5 .. 32 symbols: the kernel of lev_bitsqw_body.h (dense rings of 64 rows per symbol, `b`
    looked up by the byte) against the oracle on a sample and pair by pair against the byte-test kernels; pairs that hold a byte outside the
    alphabet -- a code no symbol has, or a symbol's code under other high bits; first / last / middle positions of either string -- are
    answered by the general kernel inside the same call.  (TA_BITSQ_WIDE=1: the kernel runs wherever it can, not only where it pays.)"""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    monkeypatch.setenv("TA_BITSQ_WIDE", "1")
    n = 200_000 + 21
    g = Dg.rng(L + Lb + k + len(alphabet))
    sym = np.frombuffer(alphabet, dtype=np.uint8)
    a = sym[g.integers(0, len(sym), size=(n, L))]
    b = sym[g.integers(0, len(sym), size=(n, Lb))]
    m = min(L, Lb)
    near = g.random(n) < 0.6
    b[near, :m] = a[near, :m]
    for row in np.nonzero(near)[0][:20000]:
        s = Dg.mutate(g, bytes(b[row]), int(g.integers(0, k + 3)), swaps=costs[3] is not None)
        s = bytes(int(sym[c % len(sym)]) if c not in sym else c for c in s)       # fold what mutate() writes into the alphabet
        s = (s + sym[g.integers(0, len(sym), size=Lb)].tobytes())[:Lb]
        b[row] = np.frombuffer(s, dtype=np.uint8)
    outside = [x for x in range(256) if x not in alphabet]
    alias = [x for x in outside if any((x & 31) == (y & 31) for y in alphabet)]  # a symbol's low five bits under other high bits
    foreign = g.choice(n, size=300, replace=False)
    for t, row in enumerate(foreign):
        byte = (alias if t % 4 == 0 and alias else outside)[int(g.integers(0, 1 << 30)) % len(alias if t % 4 == 0 and alias else outside)]
        (a if t % 2 else b)[row, [0, (Lb if t % 2 == 0 else L) - 1, m // 2][t % 3]] = byte
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    got = B.levenshtein_k_batch(sa, sb, k, costs, alphabet=alphabet).cpu().numpy().view(np.uint32)
    info = T.last_launch_info()
    assert info["kernel"] == 7 and T.last_kernel_name().startswith("lev_bitsqw_kernel<"), (info, T.last_kernel_name())
    base = B.levenshtein_k_batch(sa, sb, k, costs).cpu().numpy().view(np.uint32)
    assert T.last_launch_info()["kernel"] == 3
    assert np.array_equal(got, base), np.flatnonzero(got != base)[:10]
    ns = 20000
    assert np.array_equal(got[:ns], O.levenshtein_k_batch(O.csr_from_fixed(a[:ns]), O.csr_from_fixed(b[:ns]), k, costs))
    fr = np.sort(foreign)[:200]
    assert np.array_equal(got[fr], O.levenshtein_k_batch(O.csr_from_fixed(a[fr]), O.csr_from_fixed(b[fr]), k, costs))
    assert (base != 0xFFFFFFFF).sum() > 1000 and (base == 0xFFFFFFFF).sum() > 1000
    # a second call right away (the two fallback counters take turns), then the same alphabet through the 2-bit kernel's call path again
    again = B.levenshtein_k_batch(sa, sb, k, costs, alphabet=alphabet).cpu().numpy().view(np.uint32)
    assert np.array_equal(again, base)


def test_four_symbols_through_the_wide_alphabet_kernel(monkeypatch):
    """TA_BITSQ_WIDE=1 sends a four-letter alphabet through the 5-bit-code kernel: the same answers as the 2-bit-code kernel."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    n, L, k = 100_000, 256, 32
    g = Dg.rng(99)
    sym = np.frombuffer(b"ACGT", dtype=np.uint8)
    a = sym[g.integers(0, 4, size=(n, L))]
    b = a.copy()
    pos = g.integers(0, L, size=(n, 24))
    b[np.arange(n)[:, None], pos] = sym[g.integers(0, 4, size=(n, 24))]
    b[::3] = sym[g.integers(0, 4, size=(len(b[::3]), L))]
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    narrow = B.levenshtein_k_batch(sa, sb, k, LEV, alphabet=b"ACGT").cpu().numpy()
    assert T.last_kernel_name().startswith("lev_bitsq_kernel<")
    monkeypatch.setenv("TA_BITSQ_WIDE", "1")
    wide = B.levenshtein_k_batch(sa, sb, k, LEV, alphabet=b"ACGT").cpu().numpy()
    assert T.last_kernel_name().startswith("lev_bitsqw_kernel<")
    assert np.array_equal(narrow, wide)


def test_wide_alphabet_kernel_runs_where_it_pays():
    """The default routing of alphabets of more than four symbols (ta_levenshtein_k_batch_alphabet; profiles/r04/ab_alphabet.md): the 5-bit-code
    kernel for at most four groups of four codes at bands of 16 diagonals and more, the byte-test kernels otherwise -- the same answers."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    n, L = 60_000, 200
    g = Dg.rng(5)
    for alphabet, k, wide in ((b"ACGTN", 32, True), (b"ACGTN", 8, False), (PROTEIN, 32, False), (IUPAC, 20, False), (b"0123456789", 24, True)):
        sym = np.frombuffer(alphabet, dtype=np.uint8)
        a = sym[g.integers(0, len(sym), size=(n, L))]
        b = a.copy()
        pos = g.integers(0, L, size=(n, 12))
        b[np.arange(n)[:, None], pos] = sym[g.integers(0, len(sym), size=(n, 12))]
        sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
        got = B.levenshtein_k_batch(sa, sb, k, LEV, alphabet=alphabet).cpu().numpy().view(np.uint32)
        assert T.last_kernel_name().startswith("lev_bitsqw_kernel<") == wide, (alphabet, k, T.last_kernel_name())
        assert np.array_equal(got[:3000], O.levenshtein_k_batch(O.csr_from_fixed(a[:3000]), O.csr_from_fixed(b[:3000]), k, LEV))
