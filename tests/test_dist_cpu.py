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
