"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(tests/basic_tests.rs + doc-tests, extracted by tests/golden/extract_kats.py)."""
import pytest

import oracle_lib as O
from kat_runner import load_kats, run_kat

KATS = load_kats()


class OracleNaive:
    """Routes KATs to the *_naive* restatements (the scalar paths)."""
    supports_trace = True
    hamming = staticmethod(O.hamming_naive)
    hamming_search_with_opts = staticmethod(O.hamming_search_naive_with_opts)
    levenshtein_full = staticmethod(O.levenshtein_naive_with_opts)
    levenshtein_k_with_opts = staticmethod(O.levenshtein_naive_k_with_opts)
    levenshtein = staticmethod(O.levenshtein)
    rdamerau = staticmethod(O.rdamerau)
    levenshtein_exp = staticmethod(O.levenshtein_exp)
    rdamerau_exp = staticmethod(O.rdamerau_exp)
    levenshtein_exp_with_opts = staticmethod(O.levenshtein_exp_with_opts)
    levenshtein_search_with_opts = staticmethod(O.levenshtein_search_naive_with_opts)
    default_search_k = staticmethod(O.default_search_k)


class OraclePublic(OracleNaive):
    """Routes KATs to the public-contract restatements (dispatcher special cases included)."""
    hamming_search_with_opts = staticmethod(O.hamming_search_simd_with_opts)
    levenshtein_k_with_opts = staticmethod(O.levenshtein_simd_k_with_opts)

    @staticmethod
    def levenshtein_full(a, b, trace_on, costs):
        return O.levenshtein_simd_k_with_opts(a, b, 0xFFFFFFFF, trace_on, costs)


def _is_public(fn):
    return "naive" not in fn


@pytest.mark.parametrize("kat", KATS, ids=[k["source"].split("::")[-1] + ":" + k["fn"] for k in KATS])
def test_oracle_kat(kat):
    be = OraclePublic if _is_public(kat["fn"]) else OracleNaive
    got, want = run_kat(be, kat)
    assert got == want, kat


@pytest.mark.parametrize("kat", KATS,
                         ids=[k["source"].split("::")[-1] + ":" + k["fn"] for k in KATS])
def test_oracle_kat_cross(kat):
    """Every KAT must also hold through the *other* restatement family (naive <-> public contract),
    as the reference runs identical vectors through both."""
    if kat["fn"].startswith("hamming_search") and b"\x00" in bytes.fromhex(kat["args"][1]["hex"]):
        pytest.skip("NUL byte: public hamming_search panics by contract")
    if kat["fn"] == "hamming_search_naive_with_opts" and len(kat["args"][0]["hex"]) == 0:
        pytest.skip("empty needle differs by contract (Q3)")
    be = OracleNaive if _is_public(kat["fn"]) else OraclePublic
    got, want = run_kat(be, kat)
    assert got == want, kat


def test_kat_count():
    # 19 #[test] fns + doc-tests; guards against a silently truncated fixture
    assert len(KATS) >= 155
    fns = {k["fn"] for k in KATS}
    for must in ("levenshtein_simd_k_with_opts", "levenshtein_search_simd_with_opts", "levenshtein_exp",
                 "hamming", "hamming_search", "rdamerau", "levenshtein_naive_k_with_opts"):
        assert must in fns
