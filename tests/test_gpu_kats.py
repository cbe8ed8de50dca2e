"""-m gpu: every known-answer test of the reference (tests/golden/kats.json), tracebacks included, through the
product's HIP path (C ABI -> kernels), single-call API."""
import pytest

from kat_runner import load_kats, run_kat

pytestmark = pytest.mark.gpu
KATS = load_kats()


@pytest.mark.parametrize("kat", KATS, ids=[k["source"].split("::")[-1] + ":" + k["fn"] for k in KATS])
def test_product_kat(kat):
    from product_backend import Product
    got, want = run_kat(Product, kat)
    assert got == want, kat
