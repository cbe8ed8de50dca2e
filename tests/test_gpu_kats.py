"""-m gpu: every known-answer test of the reference (tests/golden/kats.json) through the product's
HIP path (C ABI -> kernels), single-call API.  Traceback KATs are out of scope
and must fail loudly rather than fall back."""
import pytest

from kat_runner import load_kats, needs_trace, run_kat

pytestmark = pytest.mark.gpu
KATS = load_kats()


@pytest.mark.parametrize("kat", KATS, ids=[k["source"].split("::")[-1] + ":" + k["fn"] for k in KATS])
def test_product_kat(kat):
    from product_backend import Product
    if needs_trace(kat):
        with pytest.raises(NotImplementedError):
            run_kat(Product, kat)
        return
    got, want = run_kat(Product, kat)
    assert got == want, kat
