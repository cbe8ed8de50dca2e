"""not-gpu: the multi-process layer under gloo (world_size 2 and 3): shard arithmetic, ragged all-gather and the
sharded search's halo / ownership / gather / Best-fold logic, with a CPU stand-in for the per-rank kernel."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _oracle_local_search(needle, hay_ext, k, costs, base, emit_from):
    """CPU stand-in for the HIP kernel's contract: All-mode hits of the extended shard, positions + base,
    hits with end <= emit_from (halo positions) and the end == 0 special case suppressed."""
    import oracle_lib as O
    c = (costs.mismatch_cost, costs.gap_cost, costs.start_gap_cost, costs.transpose_cost)
    hits = O.levenshtein_search_naive_with_opts(needle, hay_ext, k, O.ALL, c, False)
    rows = [(s + base, e + base, kk) for s, e, kk in hits if e > 0 and e + base > emit_from]
    return np.asarray(rows, dtype=np.int64).reshape(-1, 3)


def _worker(rank, world, port, needle, hay, k, costs, cuts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import triple_accel_amd as T
        from triple_accel_amd import dist as D
        # shard arithmetic + ragged gather
        lo, hi = D.shard_range(1003, rank, world)
        got = D.all_gather_results(torch.arange(lo, hi, dtype=torch.int32))
        assert torch.equal(got, torch.arange(1003, dtype=torch.int32))
        shard = hay[cuts[rank]:cuts[rank + 1]]
        res = {}
        for st in (T.SearchType.All, T.SearchType.Best):
            ms = D.levenshtein_search_sharded(needle, shard, k, st, T.EditCosts(*costs), local_search=_oracle_local_search)
            res[st] = [tuple(m) for m in ms]
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _run(world, needle, hay, k, costs, cuts, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, needle, hay, k, costs, cuts, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return dict(out)


@pytest.mark.parametrize("world,costs", [(2, (1, 1, 0, None)), (3, (1, 1, 0, 1)), (2, (2, 1, 2, None))])
def test_sharded_search_equals_monolithic(world, costs):
    import datagen as Dg
    import oracle_lib as O
    g = Dg.rng(123 + world)
    needle = Dg.rand_str(g, 9)
    k = 3
    hay = Dg.planted_haystack(7, needle, 3000, 70, 3)
    # uneven cuts, one of them in the middle of a planted copy, one shard shorter than the halo
    cuts = [0, 1017, 3000] if world == 2 else [0, 1017, 1025, 3000]
    res = _run(world, needle, hay, k, costs, cuts, 29500 + world * 7 + (costs[2] * 3))
    for st_name, st in (("All", O.ALL), ("Best", O.BEST)):
        want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, False)
        for r in range(world):
            key = 0 if st_name == "All" else 1
            assert res[r][key] == want, (st_name, r)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sharded_search_world_8():
    """The configuration's world size: 8 ranks, uneven cuts, two consecutive shards shorter than the halo followed by an
    EMPTY one, a planted match spanning three shards, and the globally best match (k = 0) on the last rank only."""
    import datagen as Dg
    import oracle_lib as O
    g = Dg.rng(808)
    needle = Dg.rand_str(g, 12)
    k, costs = 4, (1, 1, 0, None)                      # halo = 12 + 4 + 2 = 18
    hay = bytearray(g.integers(97, 123, size=4000, dtype=np.uint8).tobytes())

    def plant(pos, subs):
        m = bytearray(needle)
        for s in subs:
            m[s] = 32
        hay[pos:pos + len(m)] = m
    plant(100, [3]); plant(698, [1, 7]); plant(1490, [0]); plant(2195, [5, 6]); plant(3090, [2]); plant(3500, [])
    hay = bytes(hay)
    cuts = [0, 700, 705, 712, 712, 1500, 2200, 3100, 4000]      # shards of 700, 5, 7, 0, 788, 700, 900, 900 bytes
    res = _run(8, needle, hay, k, costs, cuts, _free_port())
    want_all = O.levenshtein_search_naive_with_opts(needle, hay, k, O.ALL, costs, False)
    want_best = O.levenshtein_search_naive_with_opts(needle, hay, k, O.BEST, costs, False)
    assert any(s < 700 and e > 705 for s, e, _ in want_all)     # a match that starts in shard 0 and ends in shard 2
    assert [m for m in want_best] == [(3500, 3512, 0)]
    for r in range(8):
        assert res[r][0] == want_all, r
        assert res[r][1] == want_best, r


def test_shard_range_covers_everything():
    from triple_accel_amd import dist as D
    for n in (0, 1, 7, 8, 9, 1000, 1 << 30):
        for world in (1, 2, 3, 8, 16):
            cuts = [D.shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1


# ---- pair batches sharded across ranks (distances and tracebacks): slice arithmetic + the ragged gathers, the oracle as every rank's engine
def _oracle_local_k_batch(a, b, k, costs):
    import oracle_lib as O
    c = (costs.mismatch_cost, costs.gap_cost, costs.start_gap_cost, costs.transpose_cost)
    out = np.empty(len(a), dtype=np.int32)
    for i in range(len(a)):
        d = O.levenshtein_simd_k_with_opts(a[i], b[i], k, False, c)[0]
        out[i] = -1 if d is None else d
    return out


_EDIT_CODE = {"Match": 0, "Mismatch": 1, "AGap": 2, "BGap": 3, "Transpose": 4}


def _oracle_local_trace_batch(a, b, k, costs, cap):
    import oracle_lib as O
    c = (costs.mismatch_cost, costs.gap_cost, costs.start_gap_cost, costs.transpose_cost)
    d, e, ne = np.empty(len(a), dtype=np.int32), np.zeros((len(a), cap, 2), dtype=np.int64), np.zeros(len(a), dtype=np.int32)
    for i in range(len(a)):
        wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, c)
        d[i] = -1 if wd is None else wd
        if wd is not None:
            ne[i] = len(we)
            for t, (name, cnt) in enumerate(we):
                e[i, t] = (_EDIT_CODE.get(name, name) if isinstance(name, str) else int(name), cnt)
    return d, e, ne


def _pairs_worker(rank, world, port, a, b, k, costs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import triple_accel_amd as T
        from triple_accel_amd import dist as D
        c = T.EditCosts(*costs)
        got = D.levenshtein_k_batch_sharded(a, b, k, c, local_batch=_oracle_local_k_batch)
        lo, hi = D.shard_range(len(a), rank, world)
        own = D.levenshtein_k_batch_sharded(a, b, k, c, gather=False, local_batch=_oracle_local_k_batch)
        assert own.numel() == hi - lo and torch.equal(own, got[lo:hi])
        d, e, ne = D.levenshtein_trace_batch_sharded(a, b, k, c, local_batch=_oracle_local_trace_batch)
        q.put((rank, got.numpy().copy(), d.numpy().copy(), e.numpy().copy(), ne.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 37), (3, 2), (2, 0)])
def test_sharded_pair_batches_equal_monolithic(world, n):
    import datagen as Dg
    g = Dg.rng(1234 + n)
    a, b = [], []
    for i in range(n):
        x = bytes(g.integers(97, 101, int(g.integers(0, 40)), dtype=np.uint8))
        a.append(x); b.append(Dg.mutate(g, x, 6, True) if i % 5 else bytes(g.integers(97, 101, 30, dtype=np.uint8)))
    k, costs = 5, (1, 1, 0, 1)
    import triple_accel_amd as T
    want_d = _oracle_local_k_batch(a, b, k, T.EditCosts(*costs))
    wd, we, wn = _oracle_local_trace_batch(a, b, k, T.EditCosts(*costs), 2 * k + 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_pairs_worker, args=(r, world, port, a, b, k, costs, q)) for r in range(world)]
    for p in ps:
        p.start()
    outs = [q.get(timeout=180) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, got, d, e, ne in outs:                     # the same whole-batch answer on every rank
        assert np.array_equal(got, want_d) and np.array_equal(d, wd) and np.array_equal(ne, wn)
        assert e.shape == we.shape and np.array_equal(e, we)


# ---- hamming_search over the ranks' shards + the other pair-batch forms, under gloo with the oracle as every rank's engine
def _oracle_local_hamming_search(needle, hay_ext, k, base):
    import oracle_lib as O
    hits = O.hamming_search_naive_with_opts(needle, hay_ext, k, O.ALL)
    return np.asarray([(s + base, e + base, kk) for s, e, kk in hits], dtype=np.int64).reshape(-1, 3)


def _oracle_local_exp_batch(a, b, costs):
    import oracle_lib as O
    c = (costs.mismatch_cost, costs.gap_cost, costs.start_gap_cost, costs.transpose_cost)
    return np.asarray([O.levenshtein_exp_with_opts(x, y, False, c)[0] for x, y in zip(a, b)], dtype=np.int32).reshape(-1)


def _oracle_local_hamming_batch(a, b):
    import oracle_lib as O
    out = [O.hamming_naive(x, y) for x, y in zip(a, b)]
    return np.asarray([-1 if d is None else d for d in out], dtype=np.int32).reshape(-1)


def _hsearch_worker(rank, world, port, needle, hay, k, cuts, pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import triple_accel_amd as T
        from triple_accel_amd import dist as D
        shard = hay[cuts[rank]:cuts[rank + 1]]
        res = {}
        for st in (T.SearchType.All, T.SearchType.Best):
            try:
                res[st] = [tuple(m) for m in D.hamming_search_sharded(needle, shard, k, st, local_search=_oracle_local_hamming_search)]
            except T.PanicError:
                res[st] = "panic"
        a, b = pairs
        res["exp"] = D.levenshtein_exp_batch_sharded(a, b, T.RDAMERAU_COSTS, local_batch=_oracle_local_exp_batch).numpy().copy()
        res["ham"] = D.hamming_batch_sharded(a, b, local_batch=_oracle_local_hamming_batch).numpy().copy()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _run_hsearch(world, needle, hay, k, cuts, pairs=((), ())):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_hsearch_worker, args=(r, world, port, needle, hay, k, cuts, pairs, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return dict(out)


def test_sharded_hamming_search_equals_monolithic():
    """8 ranks, uneven cuts, shards shorter than the needle, an empty shard, windows across one, two and three cuts; then a NUL byte
    in a shard that holds no window of its own: every rank panics (src/hamming.rs:463) and none hangs."""
    import datagen as Dg
    import oracle_lib as O
    g = Dg.rng(4242)
    needle = Dg.rand_str(g, 10)
    k = 3
    hay = bytearray(g.integers(97, 123, size=3000, dtype=np.uint8).tobytes())
    for pos, subs in [(50, [1]), (695, [2, 3]), (703, []), (1495, [0, 9]), (2990, [5])]:
        m = bytearray(needle)
        for s_ in subs:
            m[s_] = 35
        hay[pos:pos + 10] = m
    hay = bytes(hay)
    cuts = [0, 700, 704, 709, 709, 1500, 2200, 2995, 3000]     # shards of 700, 4, 5, 0, 791, 700, 795, 5 bytes
    a = [Dg.rand_str(g, int(g.integers(0, 30))) for _ in range(21)]
    b = [Dg.mutate(g, x, 4, True) if i % 3 else x[::-1] for i, x in enumerate(a)]
    res = _run_hsearch(8, needle, hay, k, cuts, (a, b))
    want = {0: O.hamming_search_simd_with_opts(needle, hay, k, O.ALL), 1: O.hamming_search_simd_with_opts(needle, hay, k, O.BEST)}
    assert any(s < 704 and e > 709 for s, e, _ in want[0]) and (703, 713, 0) in want[1]
    import triple_accel_amd as T
    want_exp = _oracle_local_exp_batch(a, b, T.RDAMERAU_COSTS)
    want_ham = _oracle_local_hamming_batch(a, b)
    for r in range(8):
        assert res[r][0] == want[0] and res[r][1] == want[1], r
        assert np.array_equal(res[r]["exp"], want_exp) and np.array_equal(res[r]["ham"], want_ham)
    bad = bytearray(hay)
    bad[2997] = 0                                              # inside the last shard (5 bytes: no window starts there)
    res = _run_hsearch(8, needle, bytes(bad), k, cuts)
    assert all(res[r][0] == "panic" and res[r][1] == "panic" for r in range(8))


def test_sharded_hamming_search_short_haystack_and_world_2():
    import datagen as Dg
    import oracle_lib as O
    needle = b"abcdefgh"
    res = _run_hsearch(2, needle, b"abcdef", 2, [0, 3, 6])     # needle longer than the whole haystack: empty on every rank (:455-457)
    assert all(res[r][0] == [] and res[r][1] == [] for r in range(2))
    hay = Dg.planted_haystack(5, needle, 2000, 90, 0)
    res = _run_hsearch(2, needle, hay, 2, [0, 1003, 2000])
    for r in range(2):
        assert res[r][0] == O.hamming_search_simd_with_opts(needle, hay, 2, O.ALL)
        assert res[r][1] == O.hamming_search_simd_with_opts(needle, hay, 2, O.BEST)
