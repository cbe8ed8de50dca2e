"""not-gpu: the C-ABI library loads, exports every symbol include/triple_accel_amd.h declares, its host-only
logic (cost validation, dispatcher arithmetic) matches the oracle, and compute calls FAIL LOUDLY without a GPU
(no CPU fallback)."""
import ctypes
import os
import re

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol():
    from triple_accel_amd import _native as N
    header = open(os.path.join(ROOT, "include", "triple_accel_amd.h")).read()
    declared = set(re.findall(r"\b(ta_[a-z0-9_]+)\s*\(", header))
    assert declared == set(N.ABI_SYMBOLS), declared ^ set(N.ABI_SYMBOLS)
    lib = ctypes.CDLL(N.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s


def test_costs_validation_matches_oracle():
    import triple_accel_amd as T
    for mc in (0, 1, 2, 5):
        for gc in (0, 1, 3):
            for sg in (0, 2):
                for tc in (None, 0, 1, 2, 3, 6):
                    ok = O.costs_valid((mc, gc, sg, tc))
                    try:
                        T.EditCosts(mc, gc, sg, tc)
                        got = True
                    except T.PanicError:
                        got = False
                    assert got == ok, (mc, gc, sg, tc)


def test_select_matches_oracle():
    import triple_accel_amd as T
    for la, lb in [(0, 0), (0, 5), (256, 256), (128, 100), (4096, 4096), (70000, 70000), (10, 3000)]:
        for k in (0, 1, 8, 30, 32, 60, 120, 240, 254, 255, 480, 7680, 65534, 65535, 0xFFFFFFFF):
            for c in [(1, 1, 0, None), (1, 1, 0, 1), (2, 3, 1, None), (255, 255, 255, None)]:
                assert T.levenshtein_select(la, lb, k, c) == O.levenshtein_select(la, lb, k, c), (la, lb, k, c)


def test_no_cpu_fallback():
    import torch
    import triple_accel_amd as T
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert T.device_count() == 0
    with pytest.raises(T.TripleAccelError):
        T.levenshtein(b"abc", b"abd")
    with pytest.raises(T.TripleAccelError):
        T.hamming(b"abc", b"abd")
    # argument errors are still reported before any device work, like the reference's asserts
    with pytest.raises(T.PanicError):
        T.hamming(b"ab", b"abc")


def test_product_never_touches_the_oracle():
    """The product package must not import, link or load anything under oracle/ or tests/."""
    pkg = os.path.join(ROOT, "triple_accel_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "ta_oracle" not in text and "oracle_lib" not in text and "libta_emu" not in text, f
