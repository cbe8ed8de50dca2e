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
