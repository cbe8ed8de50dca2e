"""-m gpu: the sharded search with the real HIP kernels as the per-rank search -- two processes (gloo for the tiny
control-plane gathers, both ranks on cuda:0 since the test box has one GPU) must return the monolithic oracle result.
On a multi-GPU node the same code runs one rank per GPU over RCCL (bench.py / dist.py)."""
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

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, needle, hay, k, costs, cuts, resident, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import triple_accel_amd as T
        from triple_accel_amd import dist as D
        torch.cuda.set_device(0)
        shard = hay[cuts[rank]:cuts[rank + 1]]
        if resident:                                       # the shard lives in HBM; the sharded search must not copy or re-upload it
            from triple_accel_amd import batch as B
            shard = B.haystack_tensor(shard)

            def _no_upload(*a, **kw):
                raise AssertionError("a resident shard was re-uploaded")
            B.haystack_tensor = _no_upload
        res = {}
        for st in (T.SearchType.All, T.SearchType.Best):
            ms = D.levenshtein_search_sharded(needle, shard, k, st, T.EditCosts(*costs))     # default local search: the HIP kernels
            res[int(st)] = [tuple(m) for m in ms]
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("costs,resident", [((1, 1, 0, None), False), ((1, 1, 0, None), True), ((2, 1, 1, None), True)])
def test_two_rank_sharded_search_with_hip_kernels(costs, resident):
    import datagen as Dg
    import oracle_lib as O
    g = Dg.rng(321)
    needle = Dg.rand_str(g, 24)
    k = 8
    hay = Dg.planted_haystack(11, needle, 600_000, 5000, 6)
    cuts = [0, 250_013, 600_000]                       # the cut runs through a planted copy's neighbourhood
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + costs[0] + 10 * int(resident)
    ps = [ctx.Process(target=_worker, args=(r, world, port, needle, hay, k, costs, cuts, resident, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for st in (O.ALL, O.BEST):
        want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, False)
        assert len(want) > 0
        for r in range(world):
            assert out[r][int(st)] == want, (st, r)


def _hworker(rank, world, port, needle, hay, k, cuts, resident, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import triple_accel_amd as T
        from triple_accel_amd import dist as D
        torch.cuda.set_device(0)
        shard = hay[cuts[rank]:cuts[rank + 1]]
        if resident:
            from triple_accel_amd import batch as B
            shard = B.haystack_tensor(shard)
        res = {}
        for st in (T.SearchType.All, T.SearchType.Best):
            try:
                res[int(st)] = [tuple(m) for m in D.hamming_search_sharded(needle, shard, k, st)]     # the HIP kernels on every rank
            except T.PanicError:
                res[int(st)] = "panic"
        a = [hay[i * 37:i * 37 + 20] for i in range(50)]
        b = [hay[i * 37 + 1:i * 37 + 21] for i in range(50)]
        res["ham"] = D.hamming_batch_sharded(a, b).cpu().numpy().tolist()
        res["exp"] = D.levenshtein_exp_batch_sharded(a, b).cpu().numpy().tolist()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,resident,nul", [(2, False, False), (3, True, False), (3, True, True)])
def test_sharded_hamming_search_with_hip_kernels(world, resident, nul):
    """dist.hamming_search_sharded with the real kernels as every rank's search (ranks share cuda:0, gloo for the control plane): body in
    place + tail buffer, shards shorter than the needle, a window across two cuts, the NUL rule agreed before anyone raises; and the
    sharded hamming / exp pair batches."""
    import datagen as Dg
    import oracle_lib as O
    g = Dg.rng(99 + world)
    needle = Dg.rand_str(g, 40)
    k = 6
    hay = bytearray(Dg.planted_haystack(21, needle, 400_000, 9000, 0))
    for p in range(4000, len(hay) - 100, 9000):
        for qq in g.integers(0, 40, size=4):
            hay[p + int(qq)] = 35
    cuts = [0, 200_010, 400_000] if world == 2 else [0, 200_010, 200_030, 400_000]      # (world 3: a 20-byte shard under a 40-byte needle)
    hay[200_000:200_040] = needle                                                          # a window across the cut(s)
    if nul:
        hay[200_015] = 0                                                                   # inside the 20-byte shard that holds no window
    hay = bytes(hay)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29750 + world * 3 + int(resident) + 7 * int(nul)
    ps = [ctx.Process(target=_hworker, args=(r, world, port, needle, hay, k, cuts, resident, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    a = [hay[i * 37:i * 37 + 20] for i in range(50)]
    b = [hay[i * 37 + 1:i * 37 + 21] for i in range(50)]
    want_ham = [(-1 if O.hamming_naive(x, y) is None else O.hamming_naive(x, y)) for x, y in zip(a, b)]
    want_exp = [O.levenshtein_exp_with_opts(x, y, False, (1, 1, 0, None))[0] for x, y in zip(a, b)]
    for st in (O.ALL, O.BEST):
        want = "panic" if nul else O.hamming_search_simd_with_opts(needle, hay, k, st)
        assert nul or (len(want) > 0 and any(s < 200_010 < e for s, e, _ in O.hamming_search_simd_with_opts(needle, hay, k, O.ALL)))
        for r in range(world):
            assert out[r][int(st)] == want, (st, r)
    for r in range(world):
        assert out[r]["ham"] == want_ham and out[r]["exp"] == want_exp
