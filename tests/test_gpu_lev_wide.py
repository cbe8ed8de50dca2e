"""-m gpu: the wide-band / full-matrix kernel (lev_wide.hip: row-striped, one wavefront per pair) vs the oracle."""
import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O

pytestmark = pytest.mark.gpu

COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (2, 3, 0, None), (1, 1, 2, None), (2, 1, 2, None), (2, 2, 1, 3)]


def gpu_k(a_list, b_list, k, costs):
    from triple_accel_amd import batch as B
    out = B.levenshtein_k_batch(B.Strings.from_list(a_list), B.Strings.from_list(b_list), k, costs)
    return out.cpu().numpy().view(np.uint32)


def oracle_k(a_list, b_list, k, costs):
    return O.levenshtein_k_batch(O.csr_from_list(a_list), O.csr_from_list(b_list), k, costs)


def pairs(seed, n, maxlen, kmut, swaps, minlen=0):
    g = Dg.rng(seed)
    a, b = [], []
    for i in range(n):
        x = Dg.rand_str(g, int(g.integers(minlen, maxlen + 1)))
        t = i % 4
        y = Dg.rand_str(g, int(g.integers(minlen, maxlen + 1))) if t == 0 else (x if t == 1 else Dg.mutate(g, x, kmut, swaps))
        a.append(x); b.append(y)
    return a, b


@pytest.mark.parametrize("costs", COSTS)
def test_forced_wide_small(costs, monkeypatch):
    import triple_accel_amd as T
    monkeypatch.setenv("TA_FORCE_WIDE", "1")
    a, b = pairs(5, 600, 150, 10, costs[3] is not None)
    a += [b"", b"", b"x", b"\0\0", b"ab"]; b += [b"", b"abc", b"", b"\0", b"ba"]
    for k in (0, 1, 4, 15, 60, 0xFFFFFFFF):
        got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
        assert T.last_launch_info()["kernel"] == 2
        assert np.array_equal(got, want), (k, costs, np.flatnonzero(got != want)[:10])


def test_multi_stripe_and_band_limited(monkeypatch):
    """Strings longer than one 2048-row stripe: boundary hand-over through HBM, column ranges limited by the band."""
    monkeypatch.setenv("TA_FORCE_WIDE", "1")
    g = Dg.rng(77)
    a, b = [], []
    for n in (2047, 2048, 2049, 3000, 4096, 5000, 6500):
        x = Dg.rand_str(g, n)
        a += [x, x, x, x]
        b += [Dg.mutate(g, x, 40, True), Dg.rand_str(g, n - 17), x[:n - 300], Dg.mutate(g, x, 400, True)]
    for k, costs in [(50, (1, 1, 0, None)), (500, (1, 1, 0, 1)), (3000, (1, 1, 0, None)), (0xFFFFFFFF, (1, 1, 0, 1)),
                     (900, (2, 1, 2, None)), (0xFFFFFFFF, (2, 2, 1, 3))]:
        got, want = gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)
        assert np.array_equal(got, want), (k, costs, np.flatnonzero(got != want)[:10])


def test_unforced_dispatch_to_wide():
    """A full-distance call with weighted / affine costs on strings too long for the register band picks the DP wide
    kernel by itself (the unit-cost families take the bit-parallel kernels: tests/test_gpu_lev_bits.py)."""
    import triple_accel_amd as T
    g = Dg.rng(3)
    x = Dg.rand_str(g, 6000)
    y = Dg.mutate(g, x, 100, True)
    z = Dg.rand_str(g, 5600)
    for costs in [(2, 1, 0, None), (1, 1, 1, None), (3, 2, 1, 3)]:
        c = T.EditCosts(*costs)
        for other in (y, z):
            got = T.levenshtein_simd_k_with_opts(x, other, 0xFFFFFFFF, False, c)
            assert T.last_launch_info()["kernel"] == 2
            assert got[0] == O.levenshtein_simd_k_with_opts(x, other, 0xFFFFFFFF, False, costs)[0]
    assert T.levenshtein_exp_with_opts(x, z, False, T.EditCosts(2, 1, 0, None))[0] == \
        O.levenshtein_exp_with_opts(x, z, False, (2, 1, 0, None))[0]


def test_cfg3_shape_exp_batch():
    """BASELINE cfg3 geometry (levenshtein_exp, 4 KiB pairs) at a size the oracle finishes in seconds."""
    from triple_accel_amd import batch as B
    n = 96
    ar, br = Dg.pairs_random(0x7A03, n // 2, 4096)
    am, bm = Dg.pairs_mutated_fixed(0x7A13, n // 2, 4096, 300)
    a = np.concatenate([ar, am]); b = np.concatenate([br, bm])
    out = B.levenshtein_exp_batch(B.Strings.from_fixed(a), B.Strings.from_fixed(b)).cpu().numpy().view(np.uint32)
    want = O.levenshtein_exp_batch(O.csr_from_fixed(a), O.csr_from_fixed(b))
    assert np.array_equal(out, want)
    assert (want[: n // 2] > 3840).all()        # random 4 KiB pairs need the last doubling
