#!/usr/bin/env python3
"""bench.py -- GCUPS of the banded Levenshtein hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic pairs resident in HBM.
Default workload = BASELINE.json configs[1]: levenshtein_simd_k, k = 32, 1M random 256-byte pairs,
LEVENSHTEIN_COSTS.  With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank
processes its own 1M pairs (independent units: weak scaling, no data-path collective).

One JSON line on rank 0 with `roofline` (HBM, algorithmic bytes / measured kernel time) and
`cpu_baseline` (the CPU oracle -- a restatement of the reference's scalar path -- on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (n_pairs, length, k, costs, distribution, credited cells/pair fn)
    "cfg2": dict(n=1_000_000, length=256, k=32, costs=(1, 1, 0, None), desc="levenshtein_simd_k k=32, 1M random 256B pairs, u8 cells"),
    "cfg4": dict(n=1_000_000, length=128, k=8, costs=(1, 1, 0, 1), desc="levenshtein_simd_k_with_opts RDAMERAU_COSTS k=8, 1M 128B pairs"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="override the number of pairs per GPU")
    ap.add_argument("--dist", default="random", choices=["random", "mutated"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    import torch
    import datagen as Dg
    import oracle_lib as O
    import triple_accel_amd as T
    from triple_accel_amd import batch as B

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))   # RCCL

    wl = WORKLOADS[args.workload]
    n = args.pairs or wl["n"]
    L, k, costs = wl["length"], wl["k"], wl["costs"]
    seed = 0x7A00 + int(args.workload[3:]) + 1000 * rank
    if args.dist == "random":
        a, b = Dg.pairs_random(seed, n, L)
    else:
        g = Dg.rng(seed)
        a = g.integers(33, 127, size=(n, L), dtype=np.uint8)
        b = a.copy()
        pos = g.integers(0, L, size=(n, max(1, k // 2)))
        b[np.arange(n)[:, None], pos] = 32
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    out = torch.empty(n, dtype=torch.int32, device="cuda")

    cells_pair = O.band_cells(L, L, k, costs)             # cells the scalar path visits
    bytes_pair = 2 * L + 4                                # algorithmic HBM bytes: both strings + the u32 result

    # parity gate: the timed path must equal the oracle on a sample of this very batch
    B.levenshtein_k_batch(sa, sb, k, costs, out=out)
    torch.cuda.synchronize()
    ns = min(n, 4000)
    got = out[:ns].cpu().numpy().view(np.uint32)
    want = O.levenshtein_k_batch(O.csr_from_fixed(a[:ns]), O.csr_from_fixed(b[:ns]), k, costs)
    assert np.array_equal(got, want), "parity gate failed: HIP path != oracle"
    info = T.last_launch_info()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        B.levenshtein_k_batch(sa, sb, k, costs, out=out)
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        B.levenshtein_k_batch(sa, sb, k, costs, out=out)
        ev[i + 1].record()                                # same stream as the kernel launch
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_cells = cells_pair * n * args.steps * world
    value = total_cells / elapsed / 1e9
    avg_kern_s = float(np.mean(kern_ms)) / 1e3
    achieved = bytes_pair * n / avg_kern_s / 1e9
    traffic = None
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(prof):
        try:
            traffic = json.load(open(prof)).get(args.workload, {}).get("bytes_per_launch")
        except Exception:
            traffic = None

    cpu = None
    if not args.no_cpu:
        cores = O.max_threads()
        ns_cpu = min(n, 40000 * max(1, cores // 2))       # ~10-20 s of CPU work
        ca, cb = O.csr_from_fixed(a[:ns_cpu]), O.csr_from_fixed(b[:ns_cpu])
        t1 = time.perf_counter()
        O.levenshtein_k_batch(ca, cb, k, costs, threads=cores)
        dt = time.perf_counter() - t1
        cpu = {"value": cells_pair * ns_cpu / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
               "sample": "first %d pairs of the same batch, oracle/ta_oracle.c (restated scalar levenshtein_naive_k_with_opts), "
                         "%d OpenMP threads, %.1f s" % (ns_cpu, cores, dt)}

    line = {
        "metric": "GCUPS (DP cell updates/s) for k-banded Levenshtein, 1M x 256B pairs",
        "value": value, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8 cells (reference width rule) computed in u32 lanes", "data": "synthetic",
        "config": {"workload": "%s: %s (%s bytes)" % (args.workload, wl["desc"], args.dist), "pairs_per_gpu": n,
                   "length": L, "k": k, "credited_cells_per_pair": cells_pair, "parallelism": "pairs sharded x%d" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel_ms": avg_kern_s * 1e3, "algorithmic_bytes_per_launch": bytes_pair * n,
                     "note": "integer VALU-issue-bound path (DESIGN.md section 5); HBM fraction reported as north_star asks"},
        "cpu_baseline": cpu,
        "kernel": info, "parity_checked_pairs": ns,
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
