"""-m gpu: the batch distance kernels (through the C ABI) against the CPU oracle, bit for bit."""
import os

import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O

pytestmark = pytest.mark.gpu

COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (2, 3, 0, None), (3, 1, 0, None), (1, 1, 2, None), (2, 1, 2, None),
         (2, 2, 1, 3), (5, 3, 4, 4)]


def gpu_k(a_list, b_list, k, costs):
    from triple_accel_amd import batch as B
    sa, sb = B.Strings.from_list(a_list), B.Strings.from_list(b_list)
    out = B.levenshtein_k_batch(sa, sb, k, costs)
    return out.cpu().numpy().view(np.uint32)


def oracle_k(a_list, b_list, k, costs):
    return O.levenshtein_k_batch(O.csr_from_list(a_list), O.csr_from_list(b_list), k, costs)


def ragged_pairs(seed, n, maxlen, kmut, swaps):
    g = Dg.rng(seed)
    a, b = [], []
    for i in range(n):
        x = Dg.rand_str(g, int(g.integers(0, maxlen + 1)))
        t = i % 4
        y = Dg.rand_str(g, int(g.integers(0, maxlen + 1))) if t == 0 else (x if t == 1 else Dg.mutate(g, x, kmut, swaps))
        a.append(x); b.append(y)
    return a, b


def test_dpp_and_layouts(monkeypatch):
    """Every lane layout (D, L) -- this is what pins the DPP wave_shr/wave_shl semantics on hardware."""
    a, b = ragged_pairs(11, 3000, 90, 12, True)
    for force_D, force_L in [(2, 0), (4, 0), (6, 0), (8, 3), (10, 2), (12, 0), (16, 4), (18, 1), (22, 3), (24, 0),
                             (34, 2), (66, 1), (8, 9), (2, 40), (56, 0)]:
        monkeypatch.setenv("TA_FORCE_D", str(force_D))
        monkeypatch.setenv("TA_FORCE_L", str(force_L))
        cap = force_D * (force_L if force_L else 64)
        k = min(32, max(0, (cap - 2) // 2))
        for costs in [(1, 1, 0, None), (1, 1, 0, 1), (2, 2, 1, 3)]:
            got = gpu_k(a, b, k, costs)
            want = oracle_k(a, b, k, costs)
            assert np.array_equal(got, want), (force_D, force_L, k, costs, np.flatnonzero(got != want)[:10])


@pytest.mark.parametrize("costs", COSTS)
def test_all_costs_ragged(costs):
    a, b = ragged_pairs(3, 5000, 70, 8, costs[3] is not None)
    for k in (0, 1, 3, 7, 12, 30, 64):
        got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
        assert np.array_equal(got, want), (k, costs, np.flatnonzero(got != want)[:10])


def test_cfg2_shape_100k():
    """BASELINE cfg2 geometry at a size the oracle finishes in seconds: 256 B, k = 32, strided batch."""
    from triple_accel_amd import batch as B
    import triple_accel_amd as T
    n = 100_000
    ar, br = Dg.pairs_random(0x7A02, n // 2, 256)
    am, bm = Dg.pairs_mutated_fixed(0x7A12, n // 2, 256, 32)
    a = np.concatenate([ar, am]); b = np.concatenate([br, bm])
    out = B.levenshtein_k_batch(B.Strings.from_fixed(a), B.Strings.from_fixed(b), 32).cpu().numpy().view(np.uint32)
    info = T.last_launch_info()
    assert info["kernel"] == 3 and info["cell_bits"] == 8      # unit costs, 33 diagonals: the bit-parallel band kernel
    want = O.levenshtein_k_batch(O.csr_from_fixed(a), O.csr_from_fixed(b), 32)
    assert np.array_equal(out, want)
    assert (want[: n // 2] == 0xFFFFFFFF).all() and (want[n // 2:] != 0xFFFFFFFF).mean() > 0.2


def test_cfg4_shape_100k():
    """BASELINE cfg4 geometry: 128 B, k = 8, RDAMERAU_COSTS (transposition path)."""
    from triple_accel_amd import batch as B
    n = 100_000
    am, bm = Dg.pairs_mutated_fixed(0x7A04, n, 128, 8, swaps=True)
    out = B.levenshtein_k_batch(B.Strings.from_fixed(am), B.Strings.from_fixed(bm), 8, O.RDAMERAU_COSTS)
    out = out.cpu().numpy().view(np.uint32)
    want = O.levenshtein_k_batch(O.csr_from_fixed(am), O.csr_from_fixed(bm), 8, O.RDAMERAU_COSTS)
    assert np.array_equal(out, want)
    lev = O.levenshtein_k_batch(O.csr_from_fixed(am), O.csr_from_fixed(bm), 8, O.LEVENSHTEIN_COSTS)
    assert not np.array_equal(want, lev)


def test_long_multichunk_and_edges():
    g = Dg.rng(99)
    a, b = [b"", b"", b"a", b"\0", b"\0" * 40], [b"", b"abc", b"", b"\0\0", b"\0" * 37 + b"a"]
    for n in (63, 64, 65, 127, 128, 129, 300, 500, 1000, 2000):
        x = Dg.rand_str(g, n)
        a += [x, x, x]
        b += [Dg.mutate(g, x, 20), x[: n // 2], Dg.rand_str(g, n + 5)]
    for k, costs in [(20, (1, 1, 0, None)), (25, (1, 1, 0, 1)), (40, (2, 1, 3, None)), (300, (1, 1, 0, None)),
                     (0xFFFFFFFF, (1, 1, 0, None)), (0xFFFFFFFF, (1, 1, 1, 1))]:
        got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
        assert np.array_equal(got, want), (k, costs, np.flatnonzero(got != want)[:10])


def test_full_size_properties_cfg2():
    """BASELINE cfg2 at full size (1M x 256 B): size-independent properties + a sampled oracle check.
    Mutated pairs (<= 32 unit edits) must all be Some(d <= 32), d symmetric under swapping a and b,
    and identical pairs give 0."""
    from triple_accel_amd import batch as B
    n = 1_000_000
    g = Dg.rng(0x7A22)
    a = g.integers(33, 127, size=(n, 256), dtype=np.uint8)
    b = a.copy()
    # cheap vectorised mutation: up to 16 substitutions + one block shift (<= 8 inserts/deletes) per row
    pos = g.integers(0, 256, size=(n, 16))
    b[np.arange(n)[:, None], pos] = 32
    shift = g.integers(0, 9, size=n)
    for s in range(1, 9):
        rows = np.flatnonzero(shift == s)
        b[rows, s:] = b[rows, :-s].copy()
    b[:1000] = a[:1000]
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    d_ab = B.levenshtein_k_batch(sa, sb, 32).cpu().numpy().view(np.uint32)
    d_ba = B.levenshtein_k_batch(sb, sa, 32).cpu().numpy().view(np.uint32)
    assert np.array_equal(d_ab, d_ba)
    assert (d_ab[:1000] == 0).all()
    assert (d_ab <= 32).all()
    idx = g.choice(n, size=20000, replace=False)
    want = O.levenshtein_k_batch(O.csr_from_fixed(a[idx]), O.csr_from_fixed(b[idx]), 32)
    assert np.array_equal(d_ab[idx], want)


def test_n1_equals_single_call():
    import triple_accel_amd as T
    g = Dg.rng(5)
    for _ in range(20):
        x = Dg.rand_str(g, int(g.integers(0, 50)))
        y = Dg.mutate(g, x, 5)
        assert T.levenshtein(x, y) == O.levenshtein(x, y)
        assert T.levenshtein_simd_k(x, y, 3) == O.levenshtein_simd_k_with_opts(x, y, 3)[0]
        assert T.rdamerau(x, y) == O.rdamerau(x, y)
        assert T.levenshtein_exp(x, y) == O.levenshtein_exp(x, y)


def test_exp_batch():
    from triple_accel_amd import batch as B
    g = Dg.rng(17)
    a, b = [], []
    for n in (10, 100, 300, 700):
        for _ in range(50):
            x = Dg.rand_str(g, n)
            a.append(x); b.append(Dg.mutate(g, x, n // 3) if _ % 2 else Dg.rand_str(g, n))
    for costs in [(1, 1, 0, None), (1, 1, 0, 1)]:
        out = B.levenshtein_exp_batch(B.Strings.from_list(a), B.Strings.from_list(b), costs).cpu().numpy().view(np.uint32)
        want = O.levenshtein_exp_batch(O.csr_from_list(a), O.csr_from_list(b), costs)
        assert np.array_equal(out, want)


def test_exp_batch_bag_bound(monkeypatch):
    """The doubling loop defers a pair until the threshold reaches its bag lower bound (util_kernels.hip): same distances
    with and without it, for unit and weighted costs, similar and unrelated strings in one batch."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    g = Dg.rng(171)
    a, b = [], []
    for n in (70, 200, 500, 900):
        for i in range(40):
            x = Dg.rand_str(g, n)
            y = [Dg.mutate(g, x, 3), Dg.mutate(g, x, n // 8), Dg.rand_str(g, n), Dg.rand_str(g, max(1, n - 40)), x][i % 5]
            a.append(x); b.append(y)
    sa, sb = B.Strings.from_list(a), B.Strings.from_list(b)
    ca, cb = O.csr_from_list(a), O.csr_from_list(b)
    for costs in [(1, 1, 0, None), (1, 1, 0, 1), (2, 1, 0, None), (1, 2, 0, None), (3, 2, 1, 2), (2, 3, 2, None)]:
        want = O.levenshtein_exp_batch(ca, cb, costs)
        monkeypatch.delenv("TA_EXP_NO_BOUND", raising=False)
        got = B.levenshtein_exp_batch(sa, sb, costs).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), (costs, np.flatnonzero(got != want)[:8])
        monkeypatch.setenv("TA_EXP_NO_BOUND", "1")
        got = B.levenshtein_exp_batch(sa, sb, costs).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), costs
    # fixed-length batch of unrelated strings: every pair skips the bounded rounds and goes to the unbounded pass at once
    monkeypatch.delenv("TA_EXP_NO_BOUND", raising=False)
    x, y = Dg.pairs_random(5, 300, 600)
    got = B.levenshtein_exp_batch(B.Strings.from_fixed(x), B.Strings.from_fixed(y)).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, O.levenshtein_exp_batch(O.csr_from_fixed(x), O.csr_from_fixed(y)))
    assert T.last_launch_info()["kernel"] == 4


def test_hamming_batch():
    from triple_accel_amd import batch as B
    import triple_accel_amd as T
    a, b = Dg.pairs_random(0x7A01, 10000, 1024)      # BASELINE cfg1 shape
    b[:, ::7] = a[:, ::7]
    out = B.hamming_batch(B.Strings.from_fixed(a), B.Strings.from_fixed(b)).cpu().numpy().view(np.uint32)
    assert np.array_equal(out, (a != b).sum(axis=1).astype(np.uint32))
    assert np.array_equal(out, O.hamming_batch(O.csr_from_fixed(a), O.csr_from_fixed(b)))
    la = [b"", b"abc", b"x" * 33, b"y" * 1000 + b"z"]
    lb = [b"", b"abd", b"x" * 32 + b"q", b"y" * 1001]
    out = B.hamming_batch(B.Strings.from_list(la), B.Strings.from_list(lb)).cpu().numpy().view(np.uint32)
    assert list(out) == [0, 1, 1, 1]
    out = B.hamming_batch(B.Strings.from_list([b"ab"]), B.Strings.from_list([b"abc"])).cpu().numpy().view(np.uint32)
    assert out[0] == 0xFFFFFFFF
    with pytest.raises(T.PanicError):
        T.hamming(b"ab", b"abc")


def test_transposition_forms(monkeypatch):
    """dot4-penalty form vs select form of the transposition (big mismatch costs take the select form)."""
    a, b = ragged_pairs(9, 4000, 60, 8, True)
    for costs in [(1, 1, 0, 1), (2, 2, 1, 3), (100, 90, 3, 150), (200, 130, 0, 255), (255, 255, 255, 255)]:
        for k in (3, 40, 700):
            want = oracle_k(a, b, k, costs)
            monkeypatch.delenv("TA_FORCE_TRANS_SELECT", raising=False)
            assert np.array_equal(gpu_k(a, b, k, costs), want), (costs, k)
            monkeypatch.setenv("TA_FORCE_TRANS_SELECT", "1")
            assert np.array_equal(gpu_k(a, b, k, costs), want), (costs, k)


def test_str_front_end_and_aliases():
    import triple_accel_amd as T
    assert T.levenshtein_simd_k_str("abc", "ab", 1) == 1            # src/levenshtein.rs:637-639 doc-test
    assert T.levenshtein_simd_k_str("héllo wörld", "hello world", 3) == 2
    assert T.levenshtein_simd_k_str("日本語テキスト", "日本語テスト", 1) == 1
    assert T.levenshtein_simd_k_str("日本語テキスト", "英語のテスト", 2) is None
    assert T.levenshtein_simd_k_str("日本語テキスト", "英語のテスト", 9) == 4
    assert T.levenshtein_simd_k_str("".join(chr(0x100 + i) for i in range(300)), "x", 2) is None
    assert T.hamming_words_64(b"abc", b"abd") == T.hamming_simd_movemask(b"abc", b"abd") == 1


def test_csr_without_max_len_hint_and_unaligned_views():
    """CSR batches whose longest string the library has to measure itself; blobs that start at odd addresses."""
    import torch
    from triple_accel_amd import batch as B
    a, b = ragged_pairs(21, 2000, 120, 9, False)
    sa, sb = B.Strings.from_list(a), B.Strings.from_list(b)
    sa.max_len = 0; sb.max_len = 0
    got = B.levenshtein_k_batch(sa, sb, 12).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, oracle_k(a, b, 12, (1, 1, 0, None)))
    # same data shifted by 3 bytes inside a bigger allocation (interior pointer, nothing 16-byte aligned)
    big_a = torch.zeros(sa.blob.numel() + 64, dtype=torch.uint8, device="cuda"); big_a[3:3 + sa.blob.numel()] = sa.blob
    big_b = torch.zeros(sb.blob.numel() + 64, dtype=torch.uint8, device="cuda"); big_b[5:5 + sb.blob.numel()] = sb.blob
    va = B.Strings(big_a[3:], sa.off, max_len=120); vb = B.Strings(big_b[5:], sb.off, max_len=120)
    got = B.levenshtein_k_batch(va, vb, 12).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, oracle_k(a, b, 12, (1, 1, 0, None)))


def test_u32_width_class_long_strings():
    """max_k > 65534 selects the reference's 32-bit cell class (src/levenshtein.rs:789); distances above 65535 must
    come out exact (wide kernel, 33 row stripes)."""
    import triple_accel_amd as T
    g = Dg.rng(44)
    x = Dg.rand_str(g, 33500)
    y = Dg.rand_str(g, 33400)
    costs = (3, 2, 1, None)
    assert T.levenshtein_select(len(x), len(y), 0xFFFFFFFF, costs)[2] == 32
    r = T.levenshtein_simd_k_with_opts(x, y, 0xFFFFFFFF, False, T.EditCosts(*costs))
    want = O.levenshtein_simd_k_with_opts(x, y, 0xFFFFFFFF, False, costs)
    assert r[0] == want[0] and want[0] > 65535
    assert T.last_launch_info()["cell_bits"] == 32


def _ragged_csr(seed, n, lo, hi, spread, mutated_share=0.5, kmut=24):
    """CSR batch with lengths uniform on lo..hi, b within +-spread of a; half of the pairs mutated copies, half random."""
    g = Dg.rng(seed)
    la = g.integers(lo, hi + 1, size=n)
    a_list, b_list = [], []
    for i in range(n):
        x = Dg.rand_str(g, int(la[i]))
        if g.random() < mutated_share:
            y = Dg.mutate(g, x, kmut)
        else:
            y = Dg.rand_str(g, int(np.clip(la[i] + g.integers(-spread, spread + 1), 0, hi)))
        a_list.append(x); b_list.append(y)
    return a_list, b_list


@pytest.mark.parametrize("costs,k", [((1, 1, 0, None), 32), ((1, 1, 0, 1), 8), ((2, 3, 1, None), 32), ((2, 2, 1, 3), 8)])
def test_ragged_100k_length_ordered(costs, k, monkeypatch):
    """Ragged CSR batch (lengths 32..256): the pairs are taken in length order on the device -- same answers as
    the oracle, and as the batch-order pass."""
    a, b = _ragged_csr(0x7A60 + k, 100_000, 32, 256, 40 if k == 32 else 6, kmut=24 if k == 32 else 5)
    want = oracle_k(a, b, k, costs)
    got = gpu_k(a, b, k, costs)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert (want != 0xFFFFFFFF).mean() > 0.01 and (want == 0xFFFFFFFF).mean() > 0.1        # both answers occur
    monkeypatch.setenv("TA_NO_LENGTH_ORDER", "1")
    assert np.array_equal(gpu_k(a, b, k, costs), want)


@pytest.mark.parametrize("costs", [(2, 2, 0, None), (3, 3, 0, 3), (7, 7, 0, None), (2, 2, 0, 2)])
def test_unit_costs_times_g_ride_the_bit_parallel_kernels(costs):
    """EditCosts(g, g, 0, None | Some(g)): every alignment costs g times its unit cost, so the pass runs the unit-cost kernels with k / g
    and multiplies the answers (lev_plan.h: lev_unit_scale) -- same Option as the oracle for every k, multiples of g or not; batches,
    fixed-length batches, single calls and levenshtein_exp."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    g = costs[0]
    a, b = ragged_pairs(31 + g, 6000, 120, 9, costs[3] is not None)
    for k in (0, 1, g - 1, g, g + 1, 3 * g + 1, 10 * g, 40 * g + g // 2, 0xFFFFFFFF):
        got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
        assert np.array_equal(got, want), (k, costs, np.flatnonzero(got != want)[:10])
        assert T.last_launch_info()["kernel"] in (3, 4), T.last_launch_info()      # bit-parallel kernels, not the DP band kernel
    am, bm = Dg.pairs_mutated_fixed(0x7A90 + g, 20_000, 128, 12, swaps=costs[3] is not None)
    out = B.levenshtein_k_batch(B.Strings.from_fixed(am), B.Strings.from_fixed(bm), 13 * g, costs).cpu().numpy().view(np.uint32)
    assert np.array_equal(out, O.levenshtein_k_batch(O.csr_from_fixed(am), O.csr_from_fixed(bm), 13 * g, costs))
    gg = Dg.rng(g)
    for _ in range(30):
        x = Dg.rand_str(gg, int(gg.integers(0, 300)))
        y = Dg.mutate(gg, x, 7)
        for k in (g, 5 * g + 1, 1000):
            r = T.levenshtein_simd_k_with_opts(x, y, k, False, T.EditCosts(*costs))          # None | (distance, None)
            assert (None if r is None else r[0]) == O.levenshtein_simd_k_with_opts(x, y, k, False, costs)[0]
        assert T.levenshtein_exp_with_opts(x, y, False, T.EditCosts(*costs))[0] == O.levenshtein_exp_with_opts(x, y, False, costs)[0]
    outx = B.levenshtein_exp_batch(B.Strings.from_list(a), B.Strings.from_list(b), costs).cpu().numpy().view(np.uint32)
    assert np.array_equal(outx, O.levenshtein_exp_batch(O.csr_from_list(a), O.csr_from_list(b), costs))


@pytest.mark.parametrize("costs,k", [((1, 1, 0, None), 32), ((1, 1, 0, 1), 30), ((1, 1, 0, None), 27)])
def test_ragged_vline_fetch_form(costs, k, monkeypatch):
    """The VLINE form of the bit-parallel band kernel's fetch (TA_BITS_VLINE=1: whole 128-byte lines per lane, per-lane geometry and
    alignment, one column count per pass over pairs ordered by b's exact length; lev_bits_body.h) on a ragged CSR batch -- pairs
    outside the band, near pairs, every alignment -- against the oracle, and the launcher really took it."""
    import triple_accel_amd as T
    monkeypatch.setenv("TA_BITS_VLINE", "1")
    a, b = _ragged_csr(0x7A70 + k, 60_000, 1, 300, 40, kmut=20)
    want = oracle_k(a, b, k, costs)
    got = gpu_k(a, b, k, costs)
    assert "s8v" in T.last_kernel_name(), T.last_kernel_name()
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert (want != 0xFFFFFFFF).mean() > 0.01 and (want == 0xFFFFFFFF).mean() > 0.1


def test_length_order_edge_shapes():
    """Length ordering on batches with empty strings, one giant string among short ones, and all pairs outside the band."""
    g = Dg.rng(5)
    n = 9000
    a = [Dg.rand_str(g, int(g.integers(0, 40))) for _ in range(n)]
    b = [Dg.mutate(g, x, 3) if i % 3 else b"" for i, x in enumerate(a)]
    a[1234] = Dg.rand_str(g, 70_000); b[1234] = Dg.mutate(g, a[1234], 5)
    for k in (0, 4, 100000):
        assert np.array_equal(gpu_k(a, b, k, (1, 1, 0, None)), oracle_k(a, b, k, (1, 1, 0, None))), k
    a2 = [Dg.rand_str(g, 10) for _ in range(5000)]; b2 = [Dg.rand_str(g, 30) for _ in range(5000)]
    assert np.array_equal(gpu_k(a2, b2, 5, (1, 1, 0, 1)), oracle_k(a2, b2, 5, (1, 1, 0, 1)))


def test_exp_batch_ragged_length_ordered(monkeypatch):
    """levenshtein_exp over a ragged CSR batch of 6,000 pairs: the doubling rounds take their pairs in length order; same
    distances as the oracle and as the batch-order schedule."""
    from triple_accel_amd import batch as B
    g = Dg.rng(606)
    a, b = [], []
    for i in range(6000):
        x = Dg.rand_str(g, int(g.integers(1, 400)))
        y = Dg.mutate(g, x, int(g.integers(0, 90)), True) if i % 4 else Dg.rand_str(g, int(g.integers(1, 400)))
        a.append(x); b.append(y)
    for costs in [(1, 1, 0, None), (1, 1, 0, 1), (2, 1, 1, None)]:
        want = O.levenshtein_exp_batch(O.csr_from_list(a), O.csr_from_list(b), costs)
        got = B.levenshtein_exp_batch(B.Strings.from_list(a), B.Strings.from_list(b), costs).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), (costs, np.flatnonzero(got != want)[:10])
    monkeypatch.setenv("TA_NO_LENGTH_ORDER", "1")
    got = B.levenshtein_exp_batch(B.Strings.from_list(a), B.Strings.from_list(b), (1, 1, 0, None)).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, O.levenshtein_exp_batch(O.csr_from_list(a), O.csr_from_list(b), (1, 1, 0, None)))


@pytest.mark.parametrize("costs,k,L", [((2, 3, 1, None), 32, 256), ((2, 2, 1, 3), 8, 128), ((2, 3, 0, None), 32, 256), ((3, 2, 0, None), 9, 100), ((2, 2, 2, 3), 12, 130)])
def test_band_line_form_fixed_length(costs, k, L, monkeypatch):
    """Fixed-length batches under general EditCosts in the one-lane-per-pair layout take the LINE form of the DP band kernel's fetch (every
    128-byte line of a string requested once, parked in registers; lev_band_body.h): the launcher took it, the answers are the oracle's and
    the chunk form's (TA_BAND_NO_LINE=1) -- string lengths at, below and between whole lines, different lengths of a and b."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    for la, lb in ((L, L), (L - 3, L), (L, L - 5)):
        am, bm = Dg.pairs_mutated_fixed(0x7AB0 + L + la + lb, 30_000, max(la, lb), max(2, k // 4), swaps=costs[3] is not None)
        a, b = np.ascontiguousarray(am[:, :la]), np.ascontiguousarray(bm[:, :lb])
        b[::3] = Dg.pairs_random(la + lb, len(b[::3]), lb)[1]
        want = O.levenshtein_k_batch(O.csr_from_fixed(a), O.csr_from_fixed(b), k, costs)
        got = B.levenshtein_k_batch(B.Strings.from_fixed(a), B.Strings.from_fixed(b), k, costs).cpu().numpy().view(np.uint32)
        assert "line" in T.last_kernel_name(), T.last_kernel_name()
        assert np.array_equal(got, want), (la, lb, np.flatnonzero(got != want)[:10])
        monkeypatch.setenv("TA_BAND_NO_LINE", "1")
        got2 = B.levenshtein_k_batch(B.Strings.from_fixed(a), B.Strings.from_fixed(b), k, costs).cpu().numpy().view(np.uint32)
        assert "line" not in T.last_kernel_name() and np.array_equal(got2, want)
        monkeypatch.delenv("TA_BAND_NO_LINE")


def test_exp_batch_device_driven_rounds(monkeypatch):
    """ta_levenshtein_exp_batch on batches of >= 1024 pairs enqueues its whole k schedule with the lists' lengths kept on the device (no host
    round trip between rounds): same distances as the oracle and as the host-driven loop (TA_EXP_HOST_ROUNDS=1), for every kernel family a
    round can take -- bit-parallel band (both fetch forms, two pairs per lane), row-blocked bit-parallel, DP band (cost and score form), DP
    wide -- on fixed-length and ragged batches with pairs that resolve in different rounds; and the call can be captured in a graph."""
    import torch
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    g = Dg.rng(0xE4B)
    cases = []
    # fixed length 600: mutated with 0 / 5 / 40 / 130 edits, and unrelated pairs -> rounds k = 30, 60, 120, ... and the unbounded one
    n = 3000
    x, _ = Dg.pairs_random(11, n, 600)
    y = x.copy()
    for i in range(n):
        e = (0, 5, 40, 130, -1)[i % 5]
        y[i] = np.frombuffer(Dg.mutate(g, bytes(x[i]), e, True)[:600].ljust(600, b"q"), dtype=np.uint8) if e >= 0 else Dg.pairs_random(1000 + i, 1, 600)[1][0]
    cases.append(("fixed600", B.Strings.from_fixed(x), B.Strings.from_fixed(y), O.csr_from_fixed(x), O.csr_from_fixed(y)))
    # fixed length 96 (narrow bands, two pairs per lane needs >= 262,144 pairs: one pair per lane here), similar pairs
    am, bm = Dg.pairs_mutated_fixed(0xE4C, 5000, 96, 6, swaps=True)
    cases.append(("fixed96", B.Strings.from_fixed(am), B.Strings.from_fixed(bm), O.csr_from_fixed(am), O.csr_from_fixed(bm)))
    # ragged
    a, b = [], []
    for i in range(2500):
        s = Dg.rand_str(g, int(g.integers(1, 700)))
        t = Dg.mutate(g, s, int(g.integers(0, 150)), True) if i % 3 else Dg.rand_str(g, int(g.integers(1, 700)))
        a.append(s); b.append(t)
    cases.append(("ragged", B.Strings.from_list(a), B.Strings.from_list(b), O.csr_from_list(a), O.csr_from_list(b)))
    for name, sa, sb, ca, cb in cases:
        for costs in [(1, 1, 0, None), (1, 1, 0, 1), (2, 2, 0, None), (2, 3, 1, None), (3, 2, 0, 2), (1, 2, 0, None)]:
            want = O.levenshtein_exp_batch(ca, cb, costs)
            got = B.levenshtein_exp_batch(sa, sb, costs).cpu().numpy().view(np.uint32)
            assert np.array_equal(got, want), (name, costs, np.flatnonzero(got != want)[:8])
            monkeypatch.setenv("TA_EXP_HOST_ROUNDS", "1")
            got = B.levenshtein_exp_batch(sa, sb, costs).cpu().numpy().view(np.uint32)
            monkeypatch.delenv("TA_EXP_HOST_ROUNDS")
            assert np.array_equal(got, want), (name, costs, "host rounds")
        for sw in ("TA_EXP_NO_BOUND", "TA_NO_BITS"):
            monkeypatch.setenv(sw, "1")
            got = B.levenshtein_exp_batch(sa, sb, (1, 1, 0, None)).cpu().numpy().view(np.uint32)
            monkeypatch.delenv(sw)
            assert np.array_equal(got, O.levenshtein_exp_batch(ca, cb, (1, 1, 0, None))), (name, sw)
    # the whole call inside a captured graph (fixed-length batch: nothing of it needs the host), replayed on fresh inputs of the same shape
    name, sa, sb, ca, cb = cases[0]
    out = torch.empty(sa.n, dtype=torch.int32, device="cuda")
    B.levenshtein_exp_batch(sa, sb, (1, 1, 0, None), out=out)                 # (scratch sized outside the capture)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            B.levenshtein_exp_batch(sa, sb, (1, 1, 0, None), out=out)
    out.fill_(7)
    gr.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), O.levenshtein_exp_batch(ca, cb, (1, 1, 0, None)))


@pytest.mark.parametrize("costs,k", [((2, 3, 1, None), 32), ((2, 2, 1, 3), 8), ((2, 3, 0, None), 32), ((3, 1, 0, None), 20), ((1, 2, 0, None), 12),
                                     ((4, 3, 3, 5), 40), ((3, 2, 0, 1), 9), ((2, 2, 2, 3), 12)])
def test_unit_prefilter_option(costs, k):
    """ta_set_option(TA_OPT_UNIT_PREFILTER): a weighted batch runs the unit-cost pass with k' = lev_unit_filter_k first and the DP band
    kernel prices only the pairs that pass answered -- the oracle's answers, fixed-length and ragged, pairs at / just above / far above k,
    costs whose cheapest edit is the mismatch, the gap or the transposition."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    g = Dg.rng(0x9F1 + k)
    L = 200
    am, bm = Dg.pairs_mutated_fixed(0x9F2 + k, 6000, L, max(2, k // (2 * max(costs[0], costs[1]))), swaps=costs[3] is not None)
    bm[::4] = Dg.pairs_random(0x9F3 + k, len(bm[::4]), L)[1]
    a, b = [], []
    for i in range(5000):
        s = Dg.rand_str(g, int(g.integers(1, 300)))
        t = Dg.mutate(g, s, int(g.integers(0, k // max(costs[0], costs[1]) + 3)), costs[3] is not None) if i % 3 else Dg.rand_str(g, int(g.integers(1, 300)))
        a.append(s); b.append(t)
    try:
        T.set_option(T.OPT_UNIT_PREFILTER, True)
        got = B.levenshtein_k_batch(B.Strings.from_fixed(am), B.Strings.from_fixed(bm), k, costs).cpu().numpy().view(np.uint32)
        name = T.last_kernel_name()
        got_r = B.levenshtein_k_batch(B.Strings.from_list(a), B.Strings.from_list(b), k, costs).cpu().numpy().view(np.uint32)
    finally:
        T.set_option(T.OPT_UNIT_PREFILTER, False)
    want = O.levenshtein_k_batch(O.csr_from_fixed(am), O.csr_from_fixed(bm), k, costs)
    assert np.array_equal(got, want), (costs, k, np.flatnonzero(got != want)[:10])
    assert "band" in name, name                                   # the second pass of the call is the DP band kernel's
    want_r = O.levenshtein_k_batch(O.csr_from_list(a), O.csr_from_list(b), k, costs)
    assert np.array_equal(got_r, want_r), (costs, k, np.flatnonzero(got_r != want_r)[:10])
    assert (want != 0xFFFFFFFF).sum() > 100 and (want == 0xFFFFFFFF).sum() > 100
