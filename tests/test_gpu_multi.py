"""-m gpu: the device set -- the multi-GPU split behind the C ABI (csrc/ta_multi.hip; include/triple_accel_amd.h "the device set").

The test box has ONE GPU: the set lists device 0 N times (N = 2, 3, 8), so N worker threads run the N-way partition, the chunked
pinned staging, the shard overlaps and the host-side gather exactly as eight GPUs would -- sharing one device.  Every result must equal
the one-device path's and the oracle's, bit for bit and in the same order (src/levenshtein.rs:714-720, 1911-1918, 2508-2511;
src/hamming.rs:454-475)."""
import os
import threading

import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O

pytestmark = pytest.mark.gpu

WORLDS = [2, 3, 8]


@pytest.fixture(autouse=True)
def _reset_device_set():
    from triple_accel_amd import multi as M
    keys = ["TA_MULTI_MIN_PAIRS", "TA_MULTI_MIN_HAY", "TA_MULTI_CHUNK_BYTES", "TA_MULTI_CHUNK_PAIRS", "TA_MULTI_PIECE", "TA_MULTI_DIRECT_FROM", "TA_MULTI_STAGERS_FROM"]
    saved = {k: os.environ.get(k) for k in keys}
    yield
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    M.set_devices([0])


def _tune(**kw):
    for k, v in kw.items():
        os.environ[k] = str(v)


def _ragged(seed, n, lo, hi, k):
    g = Dg.rng(seed)
    a, b = [], []
    for _ in range(n):
        s = Dg.rand_str(g, int(g.integers(lo, hi + 1)))
        a.append(s)
        b.append(Dg.mutate(g, s, k) if g.integers(0, 4) else Dg.rand_str(g, int(g.integers(lo, hi + 1))))
    return a, b


def test_device_set_roundtrip():
    from triple_accel_amd import multi as M
    M.set_devices([0, 0, 0])
    assert M.get_devices() == [0, 0, 0]
    M.set_devices(None)
    assert M.get_devices() == list(range(len(M.get_devices()))) and len(M.get_devices()) >= 1
    with pytest.raises(Exception):
        M.set_devices([99])


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("costs", [(1, 1, 0, None), (1, 1, 0, 1), (2, 3, 1, None)])
def test_host_batch_fixed_equals_oracle(world, costs):
    from triple_accel_amd import multi as M
    M.set_devices([0] * world)
    _tune(TA_MULTI_MIN_PAIRS=64, TA_MULTI_CHUNK_BYTES=96 * 1024, TA_MULTI_PIECE=16384)   # several chunks per shard, the ring wraps
    ar, br = Dg.pairs_random(5, 700, 128)
    am, bm = Dg.pairs_mutated_fixed(6, 2301, 128, 24, swaps=costs[3] is not None)
    a, b = np.concatenate([ar, am]), np.concatenate([br, bm])
    k = 24 * max(costs[0], costs[1]) + costs[2]
    got = M.levenshtein_k_batch_host(a, b, k, costs)
    want = O.levenshtein_k_batch(O.csr_from_fixed(a), O.csr_from_fixed(b), k, costs)
    assert np.array_equal(got, want)
    assert (got != 0xFFFFFFFF).sum() > 1000


@pytest.mark.parametrize("world", WORLDS)
def test_host_batch_ragged_csr_equals_oracle(world):
    from triple_accel_amd import multi as M
    M.set_devices([0] * world)
    _tune(TA_MULTI_MIN_PAIRS=32, TA_MULTI_CHUNK_BYTES=40 * 1024, TA_MULTI_PIECE=4096, TA_MULTI_DIRECT_FROM=1 << 40)      # everything through the pinned ring
    a, b = _ragged(17, 5000, 0, 200, 12)                 # empty strings included
    got = M.levenshtein_k_batch_host(a, b, 16)
    want = O.levenshtein_k_batch(O.csr_from_list(a), O.csr_from_list(b), 16)
    assert np.array_equal(got, want)
    got_e = M.levenshtein_exp_batch_host(a[:1500], b[:1500], (1, 1, 0, 1))
    want_e = O.levenshtein_exp_batch(O.csr_from_list(a[:1500]), O.csr_from_list(b[:1500]), (1, 1, 0, 1))
    assert np.array_equal(got_e, want_e)


def test_host_batch_edge_shapes():
    from triple_accel_amd import multi as M
    M.set_devices([0] * 8)
    _tune(TA_MULTI_MIN_PAIRS=1)
    assert M.levenshtein_k_batch_host([], [], 3).size == 0
    assert M.levenshtein_k_batch_host([b"kitten"], [b"sitting"], 5).tolist() == [3]            # fewer pairs than devices
    a, b = [b"", b"abc", b"", b"x" * 300, b"ab"], [b"", b"", b"abd", b"x" * 299 + b"y", b"ba"]
    assert M.levenshtein_k_batch_host(a, b, 0xFFFFFFFF).tolist() == [0, 3, 3, 1, 2]
    h = M.hamming_batch_host([b"abc", b"", b"abcd"], [b"abd", b"", b"abc"])
    assert h.tolist() == [1, 0, 0xFFFFFFFF]                                                    # a length mismatch: None
    one_big = [bytes(Dg.random_bytes(Dg.rng(3), 70000))]
    _tune(TA_MULTI_MIN_PAIRS=1, TA_MULTI_CHUNK_BYTES=4096)                                      # a pair bigger than a chunk
    assert M.levenshtein_k_batch_host(one_big + [b"a"], one_big + [b"b"], 10).tolist() == [0, 1]


@pytest.mark.parametrize("world", [1, 3])
def test_host_batch_hamming_cfg1_shape(world):
    from triple_accel_amd import multi as M
    M.set_devices([0] * world)
    _tune(TA_MULTI_MIN_PAIRS=256, TA_MULTI_STAGERS_FROM=1 << 20)                 # (two stagers per device from 1 MiB of strings; big pieces: the runtime's pageable path)
    a, b = Dg.pairs_random(9, 10_000, 1024)              # BASELINE config 1's batch
    b[::3, ::5] = a[::3, ::5]
    got = M.hamming_batch_host(a, b)
    assert np.array_equal(got, (a != b).sum(axis=1).astype(np.uint32))


@pytest.mark.parametrize("world", WORLDS)
def test_sharded_pairs_resident(world):
    from triple_accel_amd import multi as M
    M.set_devices([0] * world)
    _tune(TA_MULTI_MIN_PAIRS=16)
    a, b = Dg.pairs_mutated_fixed(21, 4099, 256, 32)
    S = M.ShardedPairs(a, b)
    assert S.n_shards == world
    ca, cb = O.csr_from_fixed(a), O.csr_from_fixed(b)
    for k, costs in [(32, (1, 1, 0, None)), (8, (1, 1, 0, 1)), (40, (2, 2, 1, 3))]:
        assert np.array_equal(S.levenshtein_k(k, costs), O.levenshtein_k_batch(ca, cb, k, costs))
    assert np.array_equal(S.levenshtein_exp(), O.levenshtein_exp_batch(ca, cb))
    assert np.array_equal(S.hamming(), (a != b).sum(axis=1).astype(np.uint32))
    assert S.time_levenshtein_k(32, steps=3) > 0
    S.close()
    # the set changes while a handle lives: the handle keeps its workers
    S2 = M.ShardedPairs(a[:500], b[:500])
    M.set_devices([0])
    assert np.array_equal(S2.levenshtein_k(32), O.levenshtein_k_batch(O.csr_from_fixed(a[:500]), O.csr_from_fixed(b[:500]), 32))
    S2.close()


def _lev_oracle(needle, hay, k, st, costs=(1, 1, 0, None)):
    return O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs)


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("costs", [(1, 1, 0, None), (1, 1, 0, 1), (2, 1, 1, None)])
def test_host_search_fans_out(world, costs):
    import triple_accel_amd as T
    from triple_accel_amd import multi as M
    M.set_devices([0] * world)
    _tune(TA_MULTI_MIN_HAY=20_000, TA_MULTI_PIECE=8192, TA_MULTI_DIRECT_FROM=(1 << 40) if world == 3 else 65536)   # (world 3: the shards through the ring)
    g = Dg.rng(77)
    needle = Dg.rand_str(g, 24)
    hay = bytearray(Dg.planted_haystack(12, needle, 300_007, 7000, 6))
    lo = 300_007 // world                                    # a copy of the needle right across the first cut
    hay[lo - 11:lo - 11 + len(needle)] = needle
    hay = bytes(hay)
    k = 8
    for st in (T.SearchType.All, T.SearchType.Best):
        got = [tuple(m) for m in T.levenshtein_search_simd_with_opts(needle, hay, k, st, T.EditCosts(*costs), False)]
        assert got == _lev_oracle(needle, hay, k, st, costs), (world, costs, st)
        assert len(got) >= 1
    # the match across the cut is there
    assert any(s <= lo < e for s, e, _ in _lev_oracle(needle, hay, k, T.SearchType.All, costs))


def test_host_search_shards_shorter_than_the_halo():
    import triple_accel_amd as T
    from triple_accel_amd import multi as M
    M.set_devices([0] * 8)
    _tune(TA_MULTI_MIN_HAY=8)                                # 8 shards of ~25 bytes under a 26-byte halo
    needle = b"abcdefghijklmnop"
    hay = b"xxabcdefghijklmnopyy" * 10
    for st in (T.SearchType.All, T.SearchType.Best):
        got = [tuple(m) for m in T.levenshtein_search_simd_with_opts(needle, hay, 8, st, T.LEVENSHTEIN_COSTS, False)]
        assert got == _lev_oracle(needle, hay, 8, st)
    # the end == 0 match (needle_len * gc + sg <= k) comes first, once
    got = [tuple(m) for m in T.levenshtein_search_simd_with_opts(b"abc", hay, 3, T.SearchType.All, T.LEVENSHTEIN_COSTS, False)]
    assert got == _lev_oracle(b"abc", hay, 3, T.SearchType.All) and got[0] == (0, 0, 3)
    # anchored searches stay on one device and keep their answer
    got = [tuple(m) for m in T.levenshtein_search_simd_with_opts(needle, hay, 4, T.SearchType.All, T.LEVENSHTEIN_COSTS, True)]
    assert got == O.levenshtein_search_naive_with_opts(needle, hay, 4, T.SearchType.All, (1, 1, 0, None), True)


@pytest.mark.parametrize("world", WORLDS)
def test_host_hamming_search_fans_out(world):
    import triple_accel_amd as T
    from triple_accel_amd import multi as M
    M.set_devices([0] * world)
    _tune(TA_MULTI_MIN_HAY=10_000)
    g = Dg.rng(5)
    for nlen, k in [(8, 2), (32, 4), (64, 10), (100, 12)]:
        needle = Dg.rand_str(g, nlen)
        hay = bytearray(Dg.planted_haystack(nlen, needle, 200_003, 3000, 0))
        for p in range(1500, len(hay) - nlen, 3000):          # substitutions only
            for q in g.integers(0, nlen, size=k):
                hay[p + int(q)] = 35
        lo = 200_003 // world
        hay[lo - nlen // 2:lo - nlen // 2 + nlen] = needle     # a window across the first cut
        hay = bytes(hay)
        for st in (T.SearchType.All, T.SearchType.Best):
            got = [tuple(m) for m in T.hamming_search_simd_with_opts(needle, hay, k, st)]
            want = O.hamming_search_simd_with_opts(needle, hay, k, st)
            assert got == want and len(got) >= 1, (world, nlen, st)
    # the SIMD contract's NUL rule holds wherever the byte sits (src/hamming.rs:463); the naive contract takes it
    for pos in (0, 200_003 // world, 200_002):
        bad = bytearray(hay)
        bad[pos] = 0
        with pytest.raises(T.PanicError):
            list(T.hamming_search_simd_with_opts(needle, bytes(bad), 3, T.SearchType.All))
        got = [tuple(m) for m in T.hamming_search_naive_with_opts(needle, bytes(bad), 12, T.SearchType.All)]
        assert got == O.hamming_search_naive_with_opts(needle, bytes(bad), 12, T.SearchType.All)


def test_host_hamming_search_tiny_shards():
    import triple_accel_amd as T
    from triple_accel_amd import multi as M
    M.set_devices([0] * 8)
    _tune(TA_MULTI_MIN_HAY=4)                                # shards shorter than the needle
    needle = b"abcdefgh"
    hay = b"abcdefgh" + b"abcdefgx" + b"zzzzzzzz" + b"abcdxfgh" + b"ab"
    for st in (T.SearchType.All, T.SearchType.Best):
        got = [tuple(m) for m in T.hamming_search_simd_with_opts(needle, hay, 2, st)]
        assert got == O.hamming_search_simd_with_opts(needle, hay, 2, st)
    bad = hay[:-1] + b"\0"                                     # a NUL in a shard that holds no window of its own
    with pytest.raises(T.PanicError):
        list(T.hamming_search_simd_with_opts(needle, bad, 2, T.SearchType.All))


@pytest.mark.parametrize("world", WORLDS)
def test_sharded_haystack_resident(world):
    import triple_accel_amd as T
    from triple_accel_amd import multi as M
    M.set_devices([0] * world)
    g = Dg.rng(99)
    needle = Dg.rand_str(g, 32)
    hay = Dg.planted_haystack(3, needle, 1 << 20, 40_000, 10)
    H = M.ShardedHaystack(hay, overlap=128)
    assert H.n_shards == world
    for k, costs in [(16, (1, 1, 0, None)), (6, (1, 1, 0, 1)), (12, (2, 3, 1, None))]:
        for st in (T.SearchType.All, T.SearchType.Best):
            got = [tuple(m) for m in H.levenshtein_search(needle, k, st, costs)]
            assert got == _lev_oracle(needle, hay, k, st, costs), (world, k, costs, st)
    for st in (T.SearchType.All, T.SearchType.Best):
        got = [tuple(m) for m in H.hamming_search(needle, 16, st)]
        assert got == O.hamming_search_simd_with_opts(needle, hay, 16, st)
    if world > 1:
        with pytest.raises(Exception):                        # needle_len + unit_k + 2 beyond the uploaded overlap
            H.levenshtein_search(Dg.rand_str(g, 120), 60)
    H.close()


def test_queue_flush_fans_out():
    import triple_accel_amd as T
    from triple_accel_amd import multi as M
    M.set_devices([0] * 3)
    _tune(TA_MULTI_MIN_PAIRS=100)
    a, b = _ragged(8, 1000, 1, 120, 9)
    got = T.levenshtein_simd_k_with_opts_many(zip(a, b), 10, T.RDAMERAU_COSTS, flush_every=700)
    want = O.levenshtein_k_batch(O.csr_from_list(a), O.csr_from_list(b), 10, (1, 1, 0, 1))
    assert [0xFFFFFFFF if d is None else d for d in got] == want.tolist()


def test_concurrent_callers_share_the_workers():
    from triple_accel_amd import multi as M
    M.set_devices([0] * 3)
    _tune(TA_MULTI_MIN_PAIRS=64, TA_MULTI_CHUNK_BYTES=64 * 1024)
    a, b = Dg.pairs_mutated_fixed(31, 3000, 128, 16)
    want = O.levenshtein_k_batch(O.csr_from_fixed(a), O.csr_from_fixed(b), 16)
    errs = []

    def run(tid):
        try:
            for _ in range(4):
                lo = 100 * tid
                got = M.levenshtein_k_batch_host(a[lo:], b[lo:], 16)
                assert np.array_equal(got, want[lo:])
        except Exception as e:                                # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_options_travel_with_the_jobs():
    import triple_accel_amd as T
    from triple_accel_amd import multi as M
    M.set_devices([0] * 2)
    _tune(TA_MULTI_MIN_PAIRS=64)
    a, b = Dg.pairs_random(41, 4096, 256)
    want = O.levenshtein_k_batch(O.csr_from_fixed(a), O.csr_from_fixed(b), 32)
    T.set_option(1, 1)                                         # TA_OPT_EARLY_OUT: same answers
    try:
        assert np.array_equal(M.levenshtein_k_batch_host(a, b, 32), want)
    finally:
        T.set_option(1, 0)


def test_many_small_calls_and_set_switches():
    """Thousands of tiny jobs and device-set switches in a row: a worker's last touch of a call's latch may come after the caller has already
    returned (round 6's fuzz run found the latch on the caller's stack: heap corruption once in ~11,000 rounds; it lives on the heap now)."""
    from triple_accel_amd import multi as M
    _tune(TA_MULTI_MIN_PAIRS=1)
    a, b = [b"kitten", b"abc", b"", b"flaw"], [b"sitting", b"abd", b"xy", b"lawn"]
    for it in range(2500):
        M.set_devices([0] * (1 + it % 4))
        assert M.levenshtein_k_batch_host(a, b, 3).tolist() == [3, 1, 2, 2]
        if it % 50 == 0:
            S = M.ShardedPairs(a, b)
            assert S.hamming().tolist() == [0xFFFFFFFF, 1, 0xFFFFFFFF, 4]
            S.close()


@pytest.mark.parametrize("world", [1, 3])
def test_host_batch_tracebacks(world):
    """ta_levenshtein_trace_batch_host: distances + edit scripts of host strings over the device set (packed runs expanded by the binding) --
    unit costs (the checkpoint route), unit costs x 2, weighted + affine + transposition (the record route), chunked -- edit for edit the oracle's."""
    from triple_accel_amd import multi as M
    M.set_devices([0] * world)
    _tune(TA_MULTI_MIN_PAIRS=64, TA_MULTI_CHUNK_BYTES=60_000)
    a, b = _ragged(23, 2500, 0, 180, 10)
    for k, costs in [(14, (1, 1, 0, None)), (12, (1, 1, 0, 1)), (24, (2, 2, 0, None)), (20, (2, 3, 1, None)), (18, (2, 2, 1, 3))]:
        d, scripts = M.levenshtein_trace_batch_host(a, b, k, costs)
        some = 0
        for i in range(0, len(a), 3):
            wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, costs)
            if wd is None:
                assert d[i] == 0xFFFFFFFF and scripts[i] == [], (i, costs)
            else:
                some += 1
                assert d[i] == wd and scripts[i] == we, (i, costs, scripts[i], we)
        assert some > 200
    fa, fb = Dg.pairs_mutated_fixed(24, 3000, 200, 16)
    d, packed, ne = M.levenshtein_trace_batch_host(fa, fb, 32, as_lists=False)
    assert packed.shape == (3000, 65) and np.array_equal(d, O.levenshtein_k_batch(O.csr_from_fixed(fa), O.csr_from_fixed(fb), 32))
    assert int(ne.max()) <= 65 and int(ne[d != 0xFFFFFFFF].min()) >= 1
