"""-m gpu: RCCL executes.  The test box has ONE GPU, so the process group has one rank -- which is enough to run every "nccl" branch of
the multi-GPU layer on the real library: init_process_group("nccl"), the all-gathers of device-resident uint8 tails and int64 match
rows inside dist.levenshtein_search_sharded, all_gather_results, and bench.py's barrier / all_reduce under the launcher
(TA_BENCH_BACKEND=nccl).  No scaling curve comes out of this; it removes dtype / device-placement surprises before an 8-GPU node
exists."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import datagen as Dg, oracle_lib as O
    import triple_accel_amd as T
    from triple_accel_amd import batch as B, dist as TD
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    g = Dg.rng(5)
    needle = Dg.rand_str(g, 24)
    hay = Dg.planted_haystack(11, needle, 600_000, 5000, 6)
    shard = B.haystack_tensor(hay)                    # resident in HBM: tails and match rows travel as CUDA tensors
    out = {}
    for costs in ((1, 1, 0, None), (2, 1, 1, None)):
        for st, ost in ((T.SearchType.All, O.ALL), (T.SearchType.Best, O.BEST)):
            got = [tuple(m) for m in TD.levenshtein_search_sharded(needle, shard, 8, st, T.EditCosts(*costs))]
            want = O.levenshtein_search_naive_with_opts(needle, hay, 8, ost, costs, False)
            assert got == want and len(want) > 0, (costs, st, len(got), len(want))
            out["%%s/%%d" %% (costs, int(st))] = len(got)
    x = TD.all_gather_results(torch.arange(7, dtype=torch.int32, device="cuda"))
    assert x.is_cuda and x.tolist() == list(range(7))
    # a pair batch's results gathered over the group
    a, b = Dg.pairs_random(3, 5000, 64)
    res = B.levenshtein_k_batch(B.Strings.from_fixed(a), B.Strings.from_fixed(b), 60)
    allr = TD.all_gather_results(res)
    assert torch.equal(allr, res)
    # the sharded pair-batch entries on the HIP path: distances and tracebacks of a ragged batch, gathered as CUDA tensors
    la, lb = [], []
    for i in range(300):
        x = Dg.rand_str(g, int(g.integers(0, 200)))
        la.append(x); lb.append(Dg.mutate(g, x, 8, True) if i %% 4 else Dg.rand_str(g, 50))
    d = TD.levenshtein_k_batch_sharded(la, lb, 10, T.RDAMERAU_COSTS)
    td, te, tn = TD.levenshtein_trace_batch_sharded(la, lb, 10, T.RDAMERAU_COSTS)
    assert d.is_cuda and te.is_cuda and te.shape == (300, 21, 2) and torch.equal(d, td)
    scripts = B.edits_to_lists(te, tn)
    for i in range(300):
        wd, we = O.levenshtein_simd_k_with_opts(la[i], lb[i], 10, True, (1, 1, 0, 1))
        assert (int(d[i]) == -1 and scripts[i] == []) if wd is None else (int(d[i]) == wd and scripts[i] == [tuple(e) for e in we]), i
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_OK " + json.dumps(out))
''')


def test_rccl_one_rank_sharded_search_and_gathers():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", _WORKER % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


@pytest.mark.parametrize("flags", [("--pairs", "20000"), ("--workload", "cfg5", "--pairs", "8")])
def test_bench_one_rank_under_the_launcher_over_rccl(flags):
    """bench.py as the driver launches it (torch.distributed.run, one rank) with TA_BENCH_BACKEND=nccl: the process group is RCCL, the
    barrier / all_reduce of the timed region run on it, cfg5 is dist.levenshtein_search_sharded (halo tails + match rows all-gathered)."""
    env = dict(os.environ, TA_BENCH_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-pmc",
           "--prewarm-ms", "20", *flags]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["config"]["backend"] == "nccl" and r["value"] > 0
    if "cfg5" in flags:
        assert "RCCL" in r["config"]["parallelism"]
