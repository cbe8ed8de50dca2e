"""-m gpu: every known-answer test of the reference (tests/golden/kats.json), tracebacks included, through the
product's HIP path (C ABI -> kernels), single-call API."""
import pytest

import oracle_lib as O
from kat_runner import load_kats, run_kat

pytestmark = pytest.mark.gpu
KATS = load_kats()


@pytest.mark.parametrize("kat", KATS, ids=[k["source"].split("::")[-1] + ":" + k["fn"] for k in KATS])
def test_product_kat(kat):
    from product_backend import Product
    got, want = run_kat(Product, kat)
    assert got == want, kat


def test_scalar_choice_where_the_simd_blend_differs():
    """Inputs on which the reference's SIMD transposition blend and its scalar routine disagree (tests/scalar_vs_simd_cases.py):
    the kernels give the scalar answer under every entry point that can reach them."""
    import triple_accel_amd as T
    from scalar_vs_simd_cases import CASES
    for a, b, costs, scalar, blend in CASES:
        C = T.EditCosts(*costs)
        assert T.levenshtein_simd_k_with_opts(a, b, 10, False, C)[0] == scalar
        assert T.levenshtein_naive_k_with_opts(a, b, 10, False, C)[0] == scalar
        assert T.rdamerau(a, b) == scalar and T.rdamerau_exp(a, b) == scalar
        d, tr = T.levenshtein_simd_k_with_opts(a, b, 10, True, C)
        want_d, want_tr = O.levenshtein_naive_k_with_opts(a, b, 10, True, costs)
        assert d == scalar == want_d and [(e.edit, e.count) for e in tr] == [(nm, c) for nm, c in want_tr]
