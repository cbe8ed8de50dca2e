"""-m gpu: the pair-sliced systolic band kernel (lev_sliced.hip: 32 pairs per register, opt-in with TA_FORCE_SLICED=1)
against the CPU oracle bit for bit, through the C ABI, and against the default bit-parallel band kernel at BASELINE size."""
import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O

def _experimental():
    try:
        import triple_accel_amd as T
        return "+experimental" in T.version()
    except Exception:
        return False


# the kernel is an experiment: it is only in libraries built with `make -C triple_accel_amd/csrc EXPERIMENTAL=1`
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _experimental(), reason="library built without EXPERIMENTAL=1")]


def _batch(seed, n, la, lb, edits):
    g = Dg.rng(seed)
    a = g.integers(33, 127, size=(n, la), dtype=np.uint8)
    b = g.integers(33, 127, size=(n, lb), dtype=np.uint8)
    m = min(la, lb)
    sim = g.random(n) < 0.7
    b[sim, :m] = a[sim, :m]
    for row in np.nonzero(sim)[0][:3000]:
        s = Dg.mutate(g, bytes(b[row]), int(g.integers(0, edits + 1)))
        s = (s + bytes(g.integers(33, 127, size=lb, dtype=np.uint8)))[:lb]
        b[row] = np.frombuffer(s, dtype=np.uint8)
    return a, b


def _run(a, b, k):
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    out = B.levenshtein_k_batch(B.Strings.from_fixed(a), B.Strings.from_fixed(b), k).cpu().numpy().view(np.uint32)
    return out, T.last_launch_info()


@pytest.mark.parametrize("n,la,lb,k,edits", [
    (1000, 256, 256, 32, 34),      # cfg2 geometry: 33 diagonals, 11 strips; a partial last wavefront
    (256 * 3, 256, 256, 32, 34),   # whole wavefronts only
    (1, 256, 256, 32, 10),         # one pair
    (5000, 200, 203, 30, 30),      # b longer: the band leans right
    (5000, 131, 127, 28, 30),      # a longer, lengths that are no multiple of 4 (unaligned dword columns)
    (3001, 64, 64, 24, 20),        # 25 diagonals, 9 strips: the narrowest group
    (2000, 300, 290, 40, 40),      # 41 diagonals, 15 strips: the widest group
    (777, 256, 256, 41, 44),
    (900, 40, 47, 36, 12),         # strings shorter than the band
    (600, 500, 512, 44, 30),       # 17 epochs
])
def test_sliced_vs_oracle(monkeypatch, n, la, lb, k, edits):
    monkeypatch.setenv("TA_FORCE_SLICED", "1")
    a, b = _batch(1000 + n + la + k, n, la, lb, edits)
    got, info = _run(a, b, k)
    assert info["kernel"] == 5, info
    want = O.levenshtein_k_batch(O.csr_from_fixed(a), O.csr_from_fixed(b), k)
    assert np.array_equal(got, want), (np.flatnonzero(got != want)[:10], got[got != want][:10], want[got != want][:10])
    assert n < 100 or ((want != 0xFFFFFFFF).any() and (want == 0xFFFFFFFF).any())


def test_sliced_is_opt_in_and_declines_what_it_cannot_do(monkeypatch):
    a, b = _batch(5, 5000, 256, 256, 30)
    monkeypatch.delenv("TA_FORCE_SLICED", raising=False)
    _, info = _run(a, b, 32)
    assert info["kernel"] == 3                                   # default: the bit-parallel band kernel
    monkeypatch.setenv("TA_FORCE_SLICED", "1")
    for k in (8, 64):                                            # band too narrow / too wide for a DPP row of strips
        got, info = _run(a, b, k)
        assert info["kernel"] == 3
        assert np.array_equal(got, O.levenshtein_k_batch(O.csr_from_fixed(a), O.csr_from_fixed(b), k))


def test_sliced_matches_bits_kernel_at_baseline_size(monkeypatch):
    """cfg2: 1M x 256 B, k = 32, mutated pairs -- two independent HIP implementations must agree on every pair."""
    n = 1_000_000
    g = Dg.rng(77)
    a = g.integers(33, 127, size=(n, 256), dtype=np.uint8)
    b = a.copy()
    pos = g.integers(0, 256, size=(n, 20))
    b[np.arange(n)[:, None], pos] = 32
    b[::3] = g.integers(33, 127, size=(len(b[::3]), 256), dtype=np.uint8)
    monkeypatch.delenv("TA_FORCE_SLICED", raising=False)
    ref, info = _run(a, b, 32)
    assert info["kernel"] == 3
    monkeypatch.setenv("TA_FORCE_SLICED", "1")
    got, info = _run(a, b, 32)
    assert info["kernel"] == 5 and info["pairs_per_wave"] == 256
    assert np.array_equal(got, ref)
    assert (ref != 0xFFFFFFFF).sum() > n // 2
