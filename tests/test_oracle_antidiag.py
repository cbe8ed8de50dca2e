"""oracle/ta_oracle_simd.c (the anti-diagonal, compiler-vectorised restatement used as bench.py's stronger cpu_baseline)
must agree bit for bit with the scalar restatement that is pinned against the reference's known-answer tests."""
import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O


def _csr(strs):
    off = np.zeros(len(strs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in strs])
    return np.frombuffer(b"".join(strs) + b"\0", dtype=np.uint8).copy(), off


def _mutate(rng, a, alpha, edits):
    b = bytearray(a)
    for _ in range(edits):
        op = int(rng.integers(0, 4))
        if op == 0 and b:
            b[int(rng.integers(0, len(b)))] = int(rng.integers(97, 97 + alpha))
        elif op == 1:
            b.insert(int(rng.integers(0, len(b) + 1)), int(rng.integers(97, 97 + alpha)))
        elif op == 2 and b:
            del b[int(rng.integers(0, len(b)))]
        elif op == 3 and len(b) > 1:
            i = int(rng.integers(0, len(b) - 1))
            b[i], b[i + 1] = b[i + 1], b[i]
    return bytes(b)


@pytest.mark.parametrize("seed", range(6))
def test_antidiag_matches_scalar_ragged(seed):
    rng = np.random.default_rng(900 + seed)
    for _ in range(40):
        alpha = int(rng.choice([2, 4, 26]))
        A, Bs = [], []
        for _ in range(int(rng.integers(1, 50))):
            a = rng.integers(97, 97 + alpha, int(rng.integers(0, 90)), dtype=np.uint8).tobytes()
            if rng.random() < 0.6:
                b = _mutate(rng, a, alpha, int(rng.integers(0, 9)))
            else:
                b = rng.integers(97, 97 + alpha, int(rng.integers(0, 90)), dtype=np.uint8).tobytes()
            A.append(a)
            Bs.append(b)
        mc = int(rng.integers(1, 4))
        gc = int(rng.integers(max(1, (mc + 1) // 2), 4))
        tc = None if rng.random() < 0.4 else int(rng.integers(1, 2 * gc + 1))
        costs = (mc, gc, 0, tc)
        if not O.costs_valid(costs):
            continue
        k = int(rng.integers(0, 45))
        want = O.levenshtein_k_batch(_csr(A), _csr(Bs), k, costs)
        got = O.levenshtein_k_batch_antidiag(_csr(A), _csr(Bs), k, costs)
        assert got is not None and np.array_equal(got, want), (costs, k)


@pytest.mark.parametrize("L,k,costs", [(256, 32, O.LEVENSHTEIN_COSTS), (128, 8, O.RDAMERAU_COSTS), (300, 600, (2, 1, 0, 2))])
def test_antidiag_matches_scalar_bench_shapes(L, k, costs):
    ar, br = Dg.pairs_random(31, 300, L)
    am, bm = Dg.pairs_mutated_fixed(32, 700, L, max(2, min(k, 40)))
    a, b = np.concatenate([ar, am]), np.concatenate([br, bm])
    ca, cb = O.csr_from_fixed(a), O.csr_from_fixed(b)
    want = O.levenshtein_k_batch(ca, cb, k, costs)
    assert np.array_equal(O.levenshtein_k_batch_antidiag(ca, cb, k, costs, threads=1), want)
    assert np.array_equal(O.levenshtein_k_batch_antidiag(ca, cb, k, costs), want)
    assert (want != 0xFFFFFFFF).any()


def test_antidiag_declines_affine_gaps_and_huge_k():
    ca, cb = _csr([b"abc"]), _csr([b"abd"])
    assert O.levenshtein_k_batch_antidiag(ca, cb, 5, (1, 1, 1, None)) is None
    assert O.levenshtein_k_batch_antidiag(ca, cb, 40000, O.LEVENSHTEIN_COSTS) is None
    assert O.levenshtein_k_batch_antidiag(ca, cb, 5, O.LEVENSHTEIN_COSTS)[0] == 1


def test_antidiag_kats():
    """The reference's own known answers for the k-bounded entries (tests/golden/kats.json) through the anti-diagonal form."""
    import json
    import os
    kats = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))["kats"]
    n = 0
    for kat in kats:
        if kat["fn"] not in ("levenshtein_simd_k_with_opts", "levenshtein_naive_k_with_opts"):
            continue
        a, b = (bytes.fromhex(x["hex"]) for x in kat["args"][:2])
        k, costs = kat["args"][2], kat["args"][4]
        if costs["start_gap"] != 0:
            continue
        got = O.levenshtein_k_batch_antidiag(_csr([a]), _csr([b]), k, costs)
        want = 0xFFFFFFFF if kat["expect"].get("none") else kat["expect"]["value"]
        assert got is not None and int(got[0]) == want, kat
        n += 1
    assert n >= 30


# ---- oracle/ta_oracle_avx2.c: hand-written AVX2, saturating u8 cells (the reference's Avx{1,2,4,8}x32x8 classes), with the
# width ladder 8 -> 16 -> 32 around it
avx2 = pytest.mark.skipif(not O.have_avx2(), reason="host CPU has no AVX2")


@avx2
@pytest.mark.parametrize("seed", range(6))
def test_ladder_matches_scalar_ragged(seed):
    rng = np.random.default_rng(1900 + seed)
    for _ in range(40):
        alpha = int(rng.choice([2, 4, 26, 255]))
        lo = 1 if alpha == 255 else 97
        A, Bs = [], []
        for _ in range(int(rng.integers(1, 50))):
            a = rng.integers(lo, lo + alpha, int(rng.integers(0, 200)), dtype=np.uint8).tobytes()
            if rng.random() < 0.6:
                b = _mutate(rng, a, min(alpha, 26), int(rng.integers(0, 12)))
            else:
                b = rng.integers(lo, lo + alpha, int(rng.integers(0, 200)), dtype=np.uint8).tobytes()
            A.append(a)
            Bs.append(b)
        mc = int(rng.integers(1, 4))
        gc = int(rng.integers(max(1, (mc + 1) // 2), 4))
        tc = None if rng.random() < 0.4 else int(rng.integers(1, 2 * gc + 1))
        costs = (mc, gc, 0, tc)
        if not O.costs_valid(costs):
            continue
        k = int(rng.choice([0, 1, 7, 8, 30, 31, 32, 33, 62, 63, 64, 100, 126, 127, 200, 254, 255, 300, 1000]))
        want = O.levenshtein_k_batch(_csr(A), _csr(Bs), k, costs)
        got = O.levenshtein_k_batch_ladder(_csr(A), _csr(Bs), k, costs)
        assert got is not None and np.array_equal(got, want), (costs, k)


@avx2
@pytest.mark.parametrize("L,k,costs,lanes", [(256, 32, O.LEVENSHTEIN_COSTS, 64), (128, 8, O.RDAMERAU_COSTS, 32),
                                             (300, 120, (1, 1, 0, 1), 128), (300, 250, (1, 1, 0, None), 256),
                                             (300, 600, (2, 1, 0, 2), 0)])
def test_ladder_bench_shapes_and_classes(L, k, costs, lanes):
    """cfg2 must run in the reference's Avx2x32x8 class (64 u8 lanes), cfg4 in Avx1x32x8 (32): levenshtein.rs:766-786."""
    ar, br = Dg.pairs_random(31, 300, L)
    am, bm = Dg.pairs_mutated_fixed(32, 700, L, max(2, min(k, 40)), swaps=costs[3] is not None)
    a, b = np.concatenate([ar, am]), np.concatenate([br, bm])
    ca, cb = O.csr_from_fixed(a), O.csr_from_fixed(b)
    want = O.levenshtein_k_batch(ca, cb, k, costs)
    got, hist = O.levenshtein_k_batch_ladder(ca, cb, k, costs, threads=1, hist=True)
    assert np.array_equal(got, want)
    assert np.array_equal(O.levenshtein_k_batch_ladder(ca, cb, k, costs), want)
    assert (want != 0xFFFFFFFF).any()
    idx = {0: 5, 32: 1, 64: 2, 128: 3, 256: 4}[lanes]
    assert hist[idx] == 1000 and sum(hist) == 1000, hist
    ref_class = O.levenshtein_select(L, L, k, costs)                   # (max_k, unit_k, cell_bits, lanes) of the reference
    assert (ref_class[2], ref_class[3]) == ((8, lanes) if lanes else (16, 0)) or lanes == 0


@avx2
def test_ladder_kats():
    import json
    import os
    kats = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))["kats"]
    n = 0
    for kat in kats:
        if kat["fn"] not in ("levenshtein_simd_k_with_opts", "levenshtein_naive_k_with_opts"):
            continue
        a, b = (bytes.fromhex(x["hex"]) for x in kat["args"][:2])
        k, costs = kat["args"][2], kat["args"][4]
        if costs["start_gap"] != 0:
            continue
        got = O.levenshtein_k_batch_ladder(_csr([a]), _csr([b]), k, costs)
        want = 0xFFFFFFFF if kat["expect"].get("none") else kat["expect"]["value"]
        assert got is not None and int(got[0]) == want, kat
        n += 1
    assert n >= 30


@avx2
def test_ladder_declines_affine_gaps():
    assert O.levenshtein_k_batch_ladder(_csr([b"abc"]), _csr([b"abd"]), 5, (1, 1, 1, None)) is None
