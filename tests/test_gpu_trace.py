"""-m gpu: traceback (trace_on = true) -- 2-bit argmin codes from the band-wavefront kernel, walk on the host --
against the oracle's scalar traceback (src/levenshtein.rs:561-606), edit for edit."""
import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O

pytestmark = pytest.mark.gpu

COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (2, 3, 0, None), (3, 1, 0, None), (1, 1, 2, None), (2, 1, 2, None), (2, 2, 1, 3)]


def prod(a, b, k, costs):
    import triple_accel_amd as T
    r = T.levenshtein_simd_k_with_opts(a, b, k, True, T.EditCosts(*costs))
    return (None, None) if r is None else (r[0], [tuple(e) for e in r[1]])


@pytest.mark.parametrize("costs", COSTS)
def test_trace_equals_scalar(costs):
    g = Dg.rng(31)
    for it in range(120):
        a = Dg.rand_str(g, int(g.integers(0, 40)))
        b = Dg.mutate(g, a, 6, costs[3] is not None) if it % 3 else Dg.rand_str(g, int(g.integers(0, 40)))
        for k in (2, 7, 30, 0xFFFFFFFF):
            want = O.levenshtein_simd_k_with_opts(a, b, k, True, costs)
            if want[0] is None:
                want = (None, None)
            assert prod(a, b, k, costs) == want, (a, b, k, costs)


def test_trace_small_alphabet_ties_and_swap():
    """Binary alphabets maximise ties; a longer first argument exercises the swap + AGap/BGap relabelling."""
    g = Dg.rng(32)
    for costs in COSTS:
        for _ in range(80):
            a = g.integers(97, 99, size=int(g.integers(0, 14)), dtype=np.uint8).tobytes()
            b = g.integers(97, 99, size=int(g.integers(0, 14)), dtype=np.uint8).tobytes()
            want = O.levenshtein_simd_k_with_opts(a, b, 0xFFFFFFFF, True, costs)
            assert prod(a, b, 0xFFFFFFFF, costs) == want, (a, b, costs)


def test_trace_longer_strings_and_exp():
    import triple_accel_amd as T
    g = Dg.rng(33)
    for n in (100, 300, 900):
        a = Dg.rand_str(g, n)
        b = Dg.mutate(g, a, n // 8, True)
        for costs in [(1, 1, 0, None), (1, 1, 0, 1), (1, 1, 2, None)]:
            want = O.levenshtein_simd_k_with_opts(a, b, n, True, costs)
            assert prod(a, b, n, costs) == want
        d, tr = T.levenshtein_exp_with_opts(a, b, True, T.LEVENSHTEIN_COSTS)
        wd, wtr = O.levenshtein_exp_with_opts(a, b, True)
        assert (d, [tuple(e) for e in tr]) == (wd, wtr)


def test_trace_beyond_the_register_band():
    """Unit-cost tracebacks whose band needs more than 64 x 66 diagonals take the row-blocked bit-parallel kernel with
    3-bit records and the host walk: same edit script as the scalar path, for several stripes, swapped roles and the
    transposition family; weighted costs on such bands stay unsupported."""
    import triple_accel_amd as T
    g = Dg.rng(34)
    for n in (5000, 9000):
        a = Dg.rand_str(g, n)
        b = Dg.mutate(g, a, n // 20, True)
        for costs in [(1, 1, 0, None), (1, 1, 0, 1)]:
            for x, y in ((a, b), (b, a)):
                want = O.levenshtein_simd_k_with_opts(x, y, 0xFFFFFFFF, True, costs)
                assert prod(x, y, 0xFFFFFFFF, costs) == want, (n, costs)
        d, tr = T.levenshtein_exp_with_opts(a, b, True, T.RDAMERAU_COSTS)
        wd, wtr = O.levenshtein_exp_with_opts(a, b, True, O.RDAMERAU_COSTS)
        assert (d, [tuple(e) for e in tr]) == (wd, wtr)
    s1 = g.integers(97, 100, size=6000, dtype=np.uint8).tobytes()
    s2 = g.integers(97, 100, size=5500, dtype=np.uint8).tobytes()
    assert prod(s1, s2, 0xFFFFFFFF, (1, 1, 0, 1)) == O.levenshtein_simd_k_with_opts(s1, s2, 0xFFFFFFFF, True, (1, 1, 0, 1))
    # weighted / affine / transposition costs on such bands: the DP wide kernel with 2-bit argmin codes
    a = Dg.rand_str(g, 4700)
    b = Dg.mutate(g, a, 120, True)
    for costs in [(2, 1, 0, None), (1, 1, 1, None), (3, 2, 1, 3), (2, 2, 0, 2)]:
        for x, y in ((a, b), (b, a)):
            want = O.levenshtein_simd_k_with_opts(x, y, 0xFFFFFFFF, True, costs)
            assert prod(x, y, 0xFFFFFFFF, costs) == want, costs
    assert prod(s1, s2, 0xFFFFFFFF, (2, 2, 0, 2)) == O.levenshtein_simd_k_with_opts(s1, s2, 0xFFFFFFFF, True, (2, 2, 0, 2))
