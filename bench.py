#!/usr/bin/env python3
"""bench.py -- GCUPS of the edit-distance hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM.
Default workload = BASELINE.json configs[1] (cfg2): levenshtein_simd_k, k = 32, 1M random 256-byte pairs,
LEVENSHTEIN_COSTS -- the configuration the metric is quoted on.  The other configs are parity-test cases;
they can be timed with --workload for DESIGN.md but are not the bench line.

Multi-GPU: `python bench.py --gpus N` launches itself as N ranks (one process per GPU, torch.distributed.run, RCCL) when
it is not already running under a launcher; under `python -m torch.distributed.run ... bench.py --gpus N` it uses the
ranks it is given.  The units are independent, so the data path has no collective:
  * --scaling weak (default, the `value` of the line): every rank its own batch of the configured size;
  * --scaling strong: the SAME batch partitioned over the ranks (dist.shard_range); at N > 1 the weak line carries this
    figure too, as `strong_scaling`, measured in the same run;
  * cfg5: the ranks' resident shards are ONE haystack (dist.levenshtein_search_sharded: halo tails all-gathered on the
    device, shard searched in place, match lists gathered -- the path's one real exchange step).
Rank 0 prints ONE JSON line carrying `roofline` (HBM: algorithmic bytes / measured kernel time, plus the VALU-issue
roofline of this integer path against both the guide's 2-cycle ceiling and the opcode-mix ceiling) and `cpu_baseline`
(the CPU oracle on a bounded sample: the hand-written AVX2 u8 anti-diagonal restatement for cfg2/cfg4 with the other
restatements and a thread-scaling line beside it, the scalar restatement elsewhere).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
PROFILE_ROUND = "r06"
ALPHABETS = {"dna": b"ACGT", "dna5": b"ACGTN", "protein": b"ACDEFGHIKLMNPQRSTVWY", "iupac": b"ACGTRYSWKMBDHVNU"}      # --dist values passed with alphabet=...


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "cfg5w", "cfg2w", "cfg4w", "cfg2l", "cfg2s", "cfg2t", "cfg2tp", "hsearch"],
                    help="cfg1..cfg5 = BASELINE.json's configs; cfg2w / cfg4w = the cfg2 / cfg4 geometry under general EditCosts "
                         "((2,3,1,None) k=32 / (2,2,1,3) k=8): the DP band-wavefront kernel; hsearch = hamming_search of a --needle-len byte needle "
                         "over a 1 GiB random shard with planted near copies (src/hamming.rs:454-554), k = needle_len / 4")
    ap.add_argument("--costs", default="2,3,1,-",
                    help="cfg5w: cfg5's geometry (32 B needle, 1 GiB shard, k = 16, Best) under these EditCosts, as mismatch,gap,start_gap,transpose "
                         "('-' = None): the unit-cost scan as a SUPERSET filter with k' = srch_filter_k + the exact kernel on the flagged blocks")
    ap.add_argument("--tcosts", default="", help="cfg2t / cfg2tp: trace under these EditCosts (mismatch,gap,start_gap,transpose; '-' = None) instead of LEVENSHTEIN_COSTS")
    ap.add_argument("--tk", type=int, default=0, help="cfg2t / cfg2tp: k (default 32)")
    ap.add_argument("--needle-len", type=int, default=32, help="hsearch: needle bytes (8 / 32: shift-add scan; > 32: SWAR kernel)")
    ap.add_argument("--pairs", type=int, default=0, help="override the number of pairs (cfg5: haystack MiB) per GPU")
    ap.add_argument("--dist", default="random", choices=["random", "mutated", "ragged", "dna", "dna5", "protein", "iupac"],
                    help="ragged: CSR batch, lengths uniform on 32..L per pair (b within +-4 of a), random bytes; cells credited pair by pair; "
                         "dna: fixed-length strings over A C G T (half of the pairs mutated copies), passed with alphabet=b'ACGT' -- the small-alphabet kernel; "
                         "dna5 / protein / iupac: the same over A C G T N / the 20 amino-acid letters / the 16 IUPAC nucleotide codes -- the kernel for alphabets of up to 32 symbols")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank its own batch; strong: the same batch partitioned over the ranks")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (default at N = 1 when rocprofv3 is on the box: two short counter passes of "
                         "the same workload -- L2 fabric-side read requests, WRITE_SIZE -- after the timed region; without them the line replays "
                         "the committed counter pass of profiles/<round>/ and says so)")
    ap.add_argument("--no-graph", action="store_true",
                    help="enqueue the K timed passes one by one; default: captured once into ONE hipGraph (where the pass is a pure kernel "
                         "launch: the fixed-length pair batches) and replayed inside the barrier-to-barrier region -- the same K passes, "
                         "without K trips through Python / ctypes / the runtime's launch path while the clock runs")
    ap.add_argument("--no-side-batch", action="store_true",
                    help="parity gate without its side batch of 2,048 mutated pairs (the counter passes use this: one more launch of the same "
                         "kernel would enter their per-pass averages)")
    ap.add_argument("--early-out", action="store_true",
                    help="ta_set_option(TA_OPT_EARLY_OUT): wavefronts stop once none of their pairs can end at or below k -- same answers, "
                         "data-dependent work; NOT the headline (the reference evaluates its whole band): the line says so in config.early_out")
    ap.add_argument("--unit-prefilter", action="store_true",
                    help="ta_set_option(TA_OPT_UNIT_PREFILTER): weighted batches (cfg2w / cfg4w / cfg2l) run the unit-cost pass with k' first and "
                         "price only its survivors -- same answers, data-dependent work; NOT a headline figure: config.unit_prefilter says so")
    ap.add_argument("--no-all-configs", action="store_true",
                    help="default run (cfg2, 1 GPU, random bytes) only: skip the four extra legs that time BASELINE.json's other configs "
                         "(cfg4, cfg5 at 1 GiB, cfg3, cfg1 -- each its own bench.py process behind its own parity gate, fewer steps, with its own "
                         "live counter passes) and report them as `all_configs` in the one JSON line")
    ap.add_argument("--single-process", action="store_true",
                    help="the multi-GPU split BEHIND the C ABI (ta_set_devices; csrc/ta_multi.hip): ONE process, host buffers in, the library's "
                         "worker threads shard the batch / the haystack over --gpus devices (a box with fewer GPUs lists device 0 that many "
                         "times).  Same JSON shape; `value` from the resident sharded handle, `end_to_end_ms` from the host-pointer entry")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed passes for this long BEFORE the W warm-up steps: the input set-up on the host leaves the GPU idle for "
                         "seconds and its clocks take longer than a handful of 0.4 ms passes to come back (0 = off)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: become N ranks (one per GPU) under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")                 # the ranks do no host-parallel work (cpu_baseline runs at N = 1 only)
    sys.exit(subprocess.call(cmd, env=env))


def host_cpu_facts():
    """What the host really offers the CPU baseline: nproc, affinity, cgroup quota, OpenMP environment."""
    facts = {"nproc": os.cpu_count()}
    try:
        facts["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        facts["affinity"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                quota = None if q < 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    facts["cgroup_cpus"] = quota
    facts["omp"] = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES", "GOMP_CPU_AFFINITY")}
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        facts["model"] = model[0] if model else None
    except Exception:
        facts["model"] = None
    return facts

def single_process(args):
    """`bench.py --gpus N --single-process`: the multi-GPU split behind the C ABI (include/triple_accel_amd.h "the device set").  ONE process;
    the strings live in host memory; the library's worker threads (one per device of the set) shard them.  `value` is measured on the
    RESIDENT sharded handle (inputs in HBM when the clock starts, as the contract asks), `end_to_end_ms` on the host-pointer entry
    (pinned staging + H2D + kernels + D2H over every device's own PCIe link)."""
    import datagen as Dg
    import oracle_lib as O
    import triple_accel_amd as T
    from triple_accel_amd import multi as M

    N = max(1, args.gpus)
    ndev = T.device_count()
    assert ndev >= 1, "bench.py needs a GPU (the product has no CPU fallback)"
    devices = [i % ndev for i in range(N)]
    M.set_devices(devices)
    wl = args.workload
    strong = args.scaling == "strong"
    LEV, RDAM = (1, 1, 0, None), (1, 1, 0, 1)
    t_wall = time.perf_counter
    if wl in ("cfg2", "cfg4", "cfg2w", "cfg4w"):
        n_cfg, L, k, costs = {"cfg2": (1_000_000, 256, 32, LEV), "cfg4": (1_000_000, 128, 8, RDAM),
                              "cfg2w": (1_000_000, 256, 32, (2, 3, 1, None)), "cfg4w": (1_000_000, 128, 8, (2, 2, 1, 3))}[wl]
        n_cfg = args.pairs or n_cfg
        n = n_cfg if strong else n_cfg * N
        a, b = Dg.pairs_random(0x7A00 + 2, n, L)
        # parity gate: a side batch of mutated pairs through the HOST entry (the fan-out, the chunked staging, the gather) ...
        am, bm = Dg.pairs_mutated_fixed(0x5EED, 4096, L, (k // max(costs[0], costs[1])) or 1, swaps=costs[3] is not None)
        want_m = O.levenshtein_k_batch(O.csr_from_fixed(am), O.csr_from_fixed(bm), k, costs)
        os.environ.setdefault("TA_MULTI_MIN_PAIRS", "256")            # (TA_TUNING only: lets the 4,096-pair side batch reach every device)
        assert np.array_equal(M.levenshtein_k_batch_host(am, bm, k, costs), want_m), "parity gate failed: host fan-out != oracle"
        os.environ.pop("TA_MULTI_MIN_PAIRS", None)
        S = M.ShardedPairs(a, b, N)
        got = S.levenshtein_k(k, costs)                               # ... and the resident handle on the timed batch
        ns = min(n, 4000)
        assert np.array_equal(got[:ns], O.levenshtein_k_batch(O.csr_from_fixed(a[:ns]), O.csr_from_fixed(b[:ns]), k, costs)), "parity gate failed"
        tail = slice(max(0, n - 1000), n)
        assert np.array_equal(got[tail], O.levenshtein_k_batch(O.csr_from_fixed(a[tail]), O.csr_from_fixed(b[tail]), k, costs)), "parity gate failed (last shard)"
        cells_total = O.band_cells(L, L, k, costs) * n
        bytes_total = (2 * L + 4) * n
        if args.prewarm_ms > 0:
            t1 = t_wall()
            while (t_wall() - t1) * 1e3 < args.prewarm_ms:
                S.time_levenshtein_k(k, costs, steps=8)
        S.time_levenshtein_k(k, costs, steps=max(1, args.warmup))
        t0 = t_wall()
        dev_ms = S.time_levenshtein_k(k, costs, steps=args.steps) / args.steps      # the slowest shard's device time (HIP events on its worker's stream)
        elapsed = t_wall() - t0                                                     # dispatch to the workers + the passes + their synchronisation
        ts = []
        for _ in range(4):
            t1 = t_wall()
            M.levenshtein_k_batch_host(a, b, k, costs)
            ts.append((t_wall() - t1) * 1e3)
        e2e_ms = float(np.median(ts[1:]))
        shards = S.n_shards
        S.close()
        desc, unit_name, units = "%s geometry: k=%d, %d random %dB pairs" % (wl, k, n, L), "pairs", n
        parity_some = int((want_m != 0xFFFFFFFF).sum())
    elif wl == "cfg5":
        mib = args.pairs or 1024
        size = (mib << 20) if strong else (mib << 20) * N
        needle = Dg.random_bytes(Dg.rng(0x7A05), 32).tobytes()
        k, costs = 16, LEV
        g = Dg.rng(0x7A05)
        hay_np = Dg.random_bytes(g, size)
        for pos in range(1 << 16, hay_np.size - 100, 1 << 20):
            mm = np.frombuffer(Dg.mutate(g, needle, 10), dtype=np.uint8)
            hay_np[pos:pos + mm.size] = mm
        H = M.ShardedHaystack(hay_np, overlap=4096, n_shards=N)
        # parity: the sharded All-mode result's records inside every shard cut's neighbourhood and at the front, against the oracle run on
        # those windows; Best == the fold over the All-mode hits
        all_hits = [tuple(m) for m in H.levenshtein_search(needle, k, T.SearchType.All, costs)]
        ns = min(size, 4 << 20)
        want = O.levenshtein_search_naive_with_opts(needle, hay_np[:ns].tobytes(), k, O.ALL, costs, False)
        assert [h for h in all_hits if h[1] <= ns] == [w for w in want if w[1] > 0], "parity gate failed: sharded search != oracle (front)"
        parity_some = len(want)
        for r in range(1, N):
            cut = size * r // N
            lo, hi = max(0, cut - (1 << 20) - 4096), min(size, cut + (1 << 20))
            w = O.levenshtein_search_naive_with_opts(needle, hay_np[lo:hi].tobytes(), k, O.ALL, costs, False)
            w = [(s0 + lo, e0 + lo, kk) for s0, e0, kk in w if e0 > 4096 or lo == 0]
            gotw = [h for h in all_hits if lo + (4096 if lo else 0) < h[1] <= hi]
            assert gotw == [x for x in w if x[1] > 0], "parity gate failed: sharded search != oracle across cut %d" % r
            parity_some += len(w)
        from triple_accel_amd import dist as TD
        best = [tuple(m) for m in H.levenshtein_search(needle, k, T.SearchType.Best, costs)]
        assert best == TD.fold_best(all_hits, k, True), "parity gate failed: Best != fold over the All-mode hits"
        run = lambda: H.levenshtein_search(needle, k, T.SearchType.Best, costs)
        t1 = t_wall()
        while (t_wall() - t1) * 1e3 < args.prewarm_ms:
            run()
        for _ in range(args.warmup):
            run()
        t0 = t_wall()
        for _ in range(args.steps):
            run()
        elapsed = t_wall() - t0
        dev_ms = None
        ts = []
        hay_bytes = hay_np.tobytes() if size <= (2 << 30) else None
        for _ in range(3 if hay_bytes else 0):
            t1 = t_wall()
            list(T.levenshtein_search_simd_with_opts(needle, hay_bytes, k, T.SearchType.Best, T.EditCosts(*costs), False))
            ts.append((t_wall() - t1) * 1e3)
        e2e_ms = float(np.median(ts[1:])) if ts else None
        shards = H.n_shards
        H.close()
        cells_total, bytes_total = 32 * size, size
        desc, unit_name, units = "levenshtein_search 32B needle over %d MiB (%d shards), k=16, Best" % (size >> 20, N), "haystack bytes", size
    else:
        sys.exit("bench.py --single-process: workloads cfg2, cfg4, cfg2w, cfg4w, cfg5")
    value = cells_total * args.steps / elapsed / 1e9
    ms_step = elapsed / args.steps * 1e3
    achieved = bytes_total / ((dev_ms if dev_ms else ms_step) / 1e3) / 1e9
    line = {
        "metric": "GCUPS (DP cell updates/s) for k-banded Levenshtein, 1M x 256B pairs" if wl == "cfg2" else "GCUPS (%s)" % wl,
        "value": value, "unit": "GCUPS", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u32 bit-vectors, 1 bit per band cell" if wl in ("cfg2", "cfg4") else "u32", "data": "synthetic",
        "config": {"workload": "%s: %s" % (wl, desc), "units_total": units, "unit": unit_name, "units_per_gpu": units // N,
                   "parallelism": "single process: ta_set_devices(%s), one worker thread + stream per entry, contiguous shards, no collective "
                                  "(host gather)" % devices, "devices": devices, "visible_devices": ndev, "shards": shards},
        "end_to_end_ms": e2e_ms,
        "end_to_end_note": "host strings in, answers out through the host-pointer entry (pinned staging ring + H2D + kernels + D2H on every device "
                           "of the set); never the headline",
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * len(set(devices)), "unit": "GB/s",
                     "frac": achieved / (HBM_PEAK_GBS * len(set(devices))), "traffic": None,
                     "device_ms_per_pass": dev_ms, "algorithmic_bytes_per_pass": bytes_total,
                     "note": "aggregate over the distinct devices of the set; the counters are collected by the one-device run"},
        "cpu_baseline": None,
        "timed_region": "%d passes back to back on every worker's stream, wall clock around dispatch + passes + synchronisation" % args.steps,
        "parity_checked_some": parity_some, "single_process": True,
    }
    print(json.dumps(line))


def main():
    args = parse_args()
    if args.single_process:
        return single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import datagen as Dg
    import oracle_lib as O
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    from triple_accel_amd import dist as TD

    if args.early_out:
        T.set_option(T.OPT_EARLY_OUT, True)
    if args.unit_prefilter:
        T.set_option(T.OPT_UNIT_PREFILTER, True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU fallback)"
    ndev = torch.cuda.device_count()
    # RCCL needs one GPU per rank; on a box with fewer GPUs than ranks (a 1-GPU dry run of the multi-rank control flow) the
    # ranks share devices and the tiny control-plane collectives go over gloo
    backend = os.environ.get("TA_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
    local = local % ndev
    torch.cuda.set_device(local)
    if world > 1:                       # one process per GPU: this rank's host entry points stay on its own device (the default set is every visible one)
        from triple_accel_amd import multi as TM
        TM.set_devices([local])
    dist = None
    # a ONE-rank run under a launcher with TA_BENCH_BACKEND set also joins a process group: every collective of the multi-rank path
    # (barrier, all_reduce, the sharded search's all-gathers) then executes on RCCL / gloo with world size 1
    dist_on = world > 1 or ("WORLD_SIZE" in os.environ and "TA_BENCH_BACKEND" in os.environ)
    if dist_on:
        import torch.distributed as dist
        import datetime
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))
        except Exception as e:                     # a clear message instead of a hang or a bare stack trace
            sys.exit("bench.py: rank %d could not join the %s process group (%s: %s).  RCCL needs one visible GPU per rank and "
                     "HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment; TA_BENCH_BACKEND=gloo runs the control plane without it."
                     % (rank, backend, type(e).__name__, e))

    wl = args.workload
    evaluated_unit = None
    cores = O.max_threads()
    quota = host_cpu_facts()["cgroup_cpus"]
    if quota:                                  # a container's CPU quota caps what OpenMP can use: more threads than that only fight
        cores = max(1, min(cores, int(quota + 0.5)))
    LEV, RDAM = (1, 1, 0, None), (1, 1, 0, 1)
    strong = args.scaling == "strong"

    # ------------------------------------------------------------------ workload set-up
    # make(seed, lo, hi) -> (run, units, parity, extra): one rank's share of a batch
    if wl in ("cfg2", "cfg4", "cfg3", "cfg1", "cfg2w", "cfg4w", "cfg2l", "cfg2s", "cfg2t", "cfg2tp"):
        n_cfg, L, k, costs = {"cfg1": (10_000, 1024, None, None), "cfg2": (1_000_000, 256, 32, LEV),
                              "cfg3": (100_000, 4096, None, LEV), "cfg4": (1_000_000, 128, 8, RDAM),
                              "cfg2w": (1_000_000, 256, 32, (2, 3, 1, None)), "cfg4w": (1_000_000, 128, 8, (2, 2, 1, 3)),
                              # cfg2's geometry under weighted LINEAR-gap costs: (2, 3, 0) -- the DP band kernel, no affine term -- and
                              # (2, 2, 0) = unit costs times two, which ride the bit-parallel kernel with k / 2
                              "cfg2l": (1_000_000, 256, 32, (2, 3, 0, None)), "cfg2s": (1_000_000, 256, 32, (2, 2, 0, None)),
                              # cfg2's geometry with trace_on = true for every pair (mutated pairs: every pair has a script):
                              # argmin codes from the DP band kernel + the walk kernel, scripts written to HBM as ta_edit runs
                              "cfg2t": (1_000_000, 256, 32, LEV),
                              # the same with PACKED records (ta_levenshtein_trace_batch_packed): 4 bytes per run, written by the walk in place
                              "cfg2tp": (1_000_000, 256, 32, LEV)}[wl]
        n_cfg = args.pairs or n_cfg
        if wl in ("cfg2t", "cfg2tp") and args.tcosts:
            cs = args.tcosts.split(",")
            costs = (int(cs[0]), int(cs[1]), int(cs[2]), None if cs[3] in ("-", "None", "") else int(cs[3]))
        if wl in ("cfg2t", "cfg2tp") and args.tk:
            k = args.tk
        ragged = args.dist == "ragged"
        assert not ragged or wl in ("cfg2", "cfg4", "cfg2w", "cfg4w", "cfg2l", "cfg2s"), "--dist ragged is a k-bounded batch distribution"
        bytes_unit = 2 * L + 4
        if wl == "cfg1":
            cells_unit = L
            desc, unit_name, dtype = "hamming() on 10K random 1KiB pairs (GPU batch kernel)", "byte pairs", "u8"
        elif wl == "cfg3":
            cells_unit = L * L                                     # the answer's work: the full matrix
            desc, unit_name, dtype = "levenshtein_exp full distance on 100K random 4KiB pairs", "pairs", "u32"
        else:
            cells_unit = O.band_cells(L, L, k, costs)              # cells the scalar banded path visits
            desc = {"cfg2": "levenshtein_simd_k k=32, 1M random 256B pairs, u8 cells",
                    "cfg4": "levenshtein_simd_k_with_opts RDAMERAU_COSTS k=8, 1M 128B pairs (transposition path)",
                    "cfg2w": "levenshtein_simd_k_with_opts EditCosts(2,3,1,None) k=32, 1M 256B pairs (general costs: DP band-wavefront kernel)",
                    "cfg4w": "levenshtein_simd_k_with_opts EditCosts(2,2,1,Some(3)) k=8, 1M 128B pairs (general costs + transposition)",
                    "cfg2l": "levenshtein_simd_k_with_opts EditCosts(2,3,0,None) k=32, 1M 256B pairs (weighted linear gaps: DP band-wavefront kernel)",
                    "cfg2s": "levenshtein_simd_k_with_opts EditCosts(2,2,0,None) k=32, 1M 256B pairs (unit costs x 2: bit-parallel kernel with k / 2)",
                    "cfg2t": "levenshtein_simd_k_with_opts k=32 trace_on=true, 1M mutated 256B pairs: distances + edit scripts (ta_levenshtein_trace_batch)",
                    "cfg2tp": "levenshtein_simd_k_with_opts k=32 trace_on=true, 1M mutated 256B pairs: distances + edit scripts as packed 4-byte runs (ta_levenshtein_trace_batch_packed)"}[wl]
            unit_name, dtype = "pairs", "u32"   # reference width class u8 (ta_levenshtein_select); the kernel computes on 1-bit cells in u32 lanes
            # cells inside the band the kernels evaluate: [min(0,delta) - t, max(0,delta) + t], t = (unit_k - |delta|) / 2
            #  -- about half of the credited reference band; reported beside the credited figure
            uk = min(max(min(k, L * max(costs[0], costs[1])) - costs[2], 0) // costs[1], 2 * L)
            evaluated_unit = sum(min(L, i + uk // 2) - max(1, i - uk // 2) + 1 for i in range(1, L + 1))

        seed0 = 0x7A00 + {"cfg1": 1, "cfg2": 2, "cfg3": 3, "cfg4": 4, "cfg2w": 12, "cfg4w": 14, "cfg2l": 22, "cfg2s": 32, "cfg2t": 42, "cfg2tp": 42}[wl]

        def gen(seed, n):
            """-> ((blob_a, off_a), (blob_b, off_b)) as numpy CSR; fixed-length distributions also carry their (n, L) arrays"""
            g = Dg.rng(seed)
            if ragged:                      # lengths uniform on 32..L, b within +-4 of a; random bytes
                la = g.integers(min(32, L), L + 1, size=n).astype(np.int64)
                lb = np.clip(la + g.integers(-4, 5, size=n), 1, L).astype(np.int64)
                csr = []
                for ln in (la, lb):
                    off = np.zeros(n + 1, dtype=np.int64)
                    np.cumsum(ln, out=off[1:])
                    blob = np.zeros(int(off[-1]) + 16, dtype=np.uint8)
                    blob[:int(off[-1])] = Dg.random_bytes(g, int(off[-1]))
                    csr.append((blob, off))
                return csr[0], csr[1], None
            if args.dist == "random" and wl not in ("cfg2t", "cfg2tp"):
                a, b = Dg.pairs_random(seed, n, L)
            elif args.dist in ALPHABETS:
                sym = np.frombuffer(ALPHABETS[args.dist], dtype=np.uint8)
                a = sym[g.integers(0, len(sym), size=(n, L))]
                b = sym[g.integers(0, len(sym), size=(n, L))]
                near = np.arange(n) % 2 == 1                      # every other pair: a copy with up to k / 2 substitutions
                b[near] = a[near]
                pos = g.integers(0, L, size=(n, max(1, (k or 64) // 2)))
                rows = np.nonzero(near)[0]
                b[rows[:, None], pos[rows]] = sym[g.integers(0, len(sym), size=(len(rows), pos.shape[1]))]
            else:
                a = g.integers(33, 127, size=(n, L), dtype=np.uint8)
                b = a.copy()
                pos = g.integers(0, L, size=(n, max(1, (k or 64) // 2)))
                b[np.arange(n)[:, None], pos] = 32
            return None, None, (a, b)

        def csr_cut(c, lo, hi):
            blob, off = c
            out = np.zeros(int(off[hi] - off[lo]) + 16, dtype=np.uint8)
            out[:-16] = blob[int(off[lo]):int(off[hi])]
            return out, (off[lo:hi + 1] - off[lo]).astype(np.uint64)

        def make(share_of_common_batch):
            """One rank's pairs: its own batch (weak) or its contiguous share of the common one (strong)."""
            ca, cb, fixed = gen(seed0 + (0 if share_of_common_batch else 1000 * rank), n_cfg)
            lo, hi = TD.shard_range(n_cfg, rank, world) if share_of_common_batch else (0, n_cfg)
            n = hi - lo
            if fixed is not None:
                a, b = fixed[0][lo:hi], fixed[1][lo:hi]
                sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
                csr = lambda l, h: (O.csr_from_fixed(a[l:h]), O.csr_from_fixed(b[l:h]))
                cells_total, bytes_total = cells_unit * n, bytes_unit * n
                host_blobs = [(sa.blob, a), (sb.blob, b)]
            else:
                ca, cb = csr_cut(ca, lo, hi), csr_cut(cb, lo, hi)
                la, lb = np.diff(ca[1].astype(np.int64)), np.diff(cb[1].astype(np.int64))
                dev = lambda c, ln: B.Strings(torch.from_numpy(c[0]).cuda(), torch.from_numpy(c[1].astype(np.int64)).cuda(),
                                              max_len=int(ln.max()) if n else 0)
                sa, sb = dev(ca, la), dev(cb, lb)
                csr = lambda l, h: (csr_cut(ca, l, h), csr_cut(cb, l, h))
                # credited pair by pair: the cells the scalar banded loop visits; a pair whose length
                # difference exceeds unit_k returns None before any cell (src/levenshtein.rs:426-428)
                memo = {}
                keys, counts = np.unique(la * 100000 + lb, return_counts=True)
                for key, cnt in zip(keys.tolist(), counts.tolist()):
                    x, y = key // 100000, key % 100000
                    uk_xy = O.levenshtein_select(x, y, k, costs)[1]          # (max_k, unit_k, cell bits, lanes)
                    memo[key] = (0 if abs(x - y) > uk_xy else O.band_cells(x, y, k, costs)) * cnt
                cells_total, bytes_total = int(sum(memo.values())), int(la.sum() + lb.sum() + 4 * n)
                host_blobs = [(sa.blob, ca[0]), (sb.blob, cb[0]), (sa.off, ca[1].astype(np.int64)), (sb.off, cb[1].astype(np.int64))]
            out = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")[:n]
            if wl in ("cfg2t", "cfg2tp"):
                cap = 2 * k + 1
                ne = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")[:n]
                if wl == "cfg2t":
                    ed = torch.empty((max(n, 1), cap, 2), dtype=torch.int64, device="cuda")[:n]
                    run = lambda: B.levenshtein_trace_batch(sa, sb, k, costs, cap=cap, out=out, edits=ed, n_edits=ne)
                else:
                    ed = torch.empty((max(n, 1), cap), dtype=torch.int32, device="cuda")[:n]
                    run = lambda: B.levenshtein_trace_batch_packed(sa, sb, k, costs, cap=cap, out=out, packed=ed, n_edits=ne)
                oracle = lambda l, h, th: O.levenshtein_k_batch(*csr(l, h), k, costs, threads=th)
            elif wl == "cfg1":
                run = lambda: B.hamming_batch(sa, sb, out=out)
                oracle = lambda l, h, th: O.hamming_batch(*csr(l, h), threads=th)
            elif wl == "cfg3":
                run = lambda: B.levenshtein_exp_batch(sa, sb, costs, out=out)
                oracle = lambda l, h, th: O.levenshtein_exp_batch(*csr(l, h), costs, threads=th)
            else:
                alphabet = ALPHABETS.get(args.dist)
                run = lambda: B.levenshtein_k_batch(sa, sb, k, costs, out=out, alphabet=alphabet)
                out_b = torch.empty_like(out)                 # (the overlapped figure: consecutive passes on two streams write two buffers)
                run_alt = lambda: B.levenshtein_k_batch(sa, sb, k, costs, out=out_b, alphabet=alphabet)
                oracle = lambda l, h, th: O.levenshtein_k_batch(*csr(l, h), k, costs, threads=th)
            if n == 0:
                run = lambda: None

            def parity():
                some = side_n = 0
                if k is not None and wl != "cfg3" and fixed is not None and not args.no_side_batch:
                    # Random pairs are all None at these k: the gate below then compares sentinels.  A SIDE batch of mutated pairs of the
                    # same geometry (real distances; swaps for the transposition family) goes through the same entry point, costs, k
                    # and alphabet -- the same kernel selection rule -- and is compared answer by answer.  (It runs BEFORE the timed
                    # batch's pass so that last_launch_info / last_kernel_name describe the timed batch.)
                    side_n = 2048
                    am, bm = Dg.pairs_mutated_fixed(seed0 + 0x5EED, side_n, L, (k // max(costs[0], costs[1], 1)) or 1, swaps=costs[3] is not None)
                    if args.dist in ALPHABETS:                       # keep the side batch inside the alphabet the kernel was promised
                        sym = np.frombuffer(ALPHABETS[args.dist], dtype=np.uint8)
                        am, bm = sym[am % len(sym)], sym[bm % len(sym)]
                    side = torch.empty(side_n, dtype=torch.int32, device="cuda")
                    if wl in ("cfg2t", "cfg2tp"):
                        B.levenshtein_trace_batch(B.Strings.from_fixed(am), B.Strings.from_fixed(bm), k, costs, out=side)
                    else:
                        B.levenshtein_k_batch(B.Strings.from_fixed(am), B.Strings.from_fixed(bm), k, costs, out=side, alphabet=ALPHABETS.get(args.dist))
                    want2 = O.levenshtein_k_batch(O.csr_from_fixed(am), O.csr_from_fixed(bm), k, costs, threads=cores)
                    assert np.array_equal(side.cpu().numpy().view(np.uint32), want2), "parity gate failed on the mutated side batch: HIP path != oracle"
                    some += int((want2 != 0xFFFFFFFF).sum())
                run(); torch.cuda.synchronize()
                ns = min(n, 4000 if wl != "cfg3" else 48)
                got = out[:ns].cpu().numpy().view(np.uint32)
                want = oracle(0, ns, cores)
                assert np.array_equal(got, want), "parity gate failed: HIP path != oracle"
                extra_t["parity_some"] = some + int((want != 0xFFFFFFFF).sum())      # (cfg1 / cfg3: every answer is a number)
                extra_t["parity_side"] = side_n
                if wl in ("cfg2t", "cfg2tp"):               # the scripts, edit for edit, against the scalar traceback; their bytes
                    scripts = B.edits_to_lists(ed[:600], ne[:600]) if wl == "cfg2t" else B.packed_to_lists(ed[:600], ne[:600])
                    for i in range(min(n, 600)):
                        wd, we = O.levenshtein_simd_k_with_opts(fixed[0][lo + i].tobytes(), fixed[1][lo + i].tobytes(), k, True, costs)
                        assert scripts[i] == (we if wd is not None else []), "parity gate failed: edit script != oracle's traceback"
                    extra_t["runs_total"] = int(ne.to(torch.int64).sum().item())
                return ns

            def end_to_end(reps=5):
                """host buffers in, answers out: pinned H2D of the batch + the pass + D2H of the results; ms (median)"""
                pins = [(dst, torch.from_numpy(np.ascontiguousarray(src).reshape(-1)).pin_memory()) for dst, src in host_blobs]
                out_h = torch.empty(max(n, 1), dtype=torch.int32).pin_memory()
                ts = []
                for _ in range(reps + 1):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for dst, src in pins:
                        dst[: src.numel()].copy_(src, non_blocking=True)
                    run()
                    out_h[:n].copy_(out, non_blocking=True)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1) * 1e3)
                return float(np.median(ts[1:]))
            extra_t = {"csr": csr, "oracle": oracle, "cells_total": cells_total, "bytes_total": bytes_total, "end_to_end": end_to_end}
            if n and wl in ("cfg2", "cfg4", "cfg2w", "cfg4w", "cfg2l", "cfg2s"):
                extra_t["run_alt"], extra_t["out_pair"] = run_alt, (out, out_b)
            return run, n, parity, extra_t
    elif wl == "hsearch":
        # hamming_search over a haystack shard resident in HBM: every offset's mismatch count against the needle, reported when <= k.
        # Algorithmic bytes: the haystack once (+ the few hit records); cells: needle_len byte comparisons per offset.
        mib = args.pairs or 1024
        nlen = args.needle_len
        needle = Dg.random_bytes(Dg.rng(0x7A06), nlen).tobytes()
        needle = bytes(c or 1 for c in needle)                      # (no NUL bytes: the reference panics on them)
        k = max(1, nlen // 4)
        cells_unit, bytes_unit = nlen, 1
        desc = "hamming_search %dB needle over a %d MiB random haystack shard per GPU, k=%d" % (nlen, mib, k)
        unit_name, dtype = "haystack bytes", "u8 counters (byte compares)"

        def make(share_of_common_batch):
            size = mib << 20
            g = Dg.rng(0x7A06 + 1000 * rank)
            hay_np = Dg.random_bytes(g, size)
            hay_np[hay_np == 0] = 1
            nd = np.frombuffer(needle, dtype=np.uint8)
            for pos in range(1 << 16, hay_np.size - 2 * nlen, 1 << 20):     # ~1 planted copy per MiB with k/2 substitutions
                hay_np[pos:pos + nlen] = nd
                for q in g.integers(0, nlen, size=max(1, k // 2)):
                    hay_np[pos + int(q)] = 7
            hay = B.haystack_tensor(hay_np)
            holder = {}

            def run():
                holder["hits"] = B.hamming_search_dev(needle, hay, k)

            def parity():
                run(); torch.cuda.synchronize()
                ns = min(hay_np.size, 2 << 20)
                want = O.hamming_search_naive_with_opts(needle, hay_np[:ns].tobytes(), k, O.ALL)
                got = [tuple(int(v) for v in r) for r in holder["hits"] if r[1] <= ns]
                assert got == [tuple(w) for w in want] and len(want) > 0, "parity gate failed: HIP hamming_search != oracle"
                return ns

            def end_to_end(reps=3):
                pin = torch.from_numpy(hay_np).pin_memory()
                ts = []
                for _ in range(reps + 1):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    hay[0][: pin.numel()].copy_(pin, non_blocking=True)
                    run()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1) * 1e3)
                return float(np.median(ts[1:]))
            return run, hay_np.size, parity, {"hay_np": hay_np, "cells_total": cells_unit * hay_np.size, "bytes_total": bytes_unit * hay_np.size,
                                              "end_to_end": end_to_end}
    else:   # cfg5: levenshtein_search, 32 B needle over a 1 GiB random shard per GPU
        mib = args.pairs or 1024
        needle = Dg.random_bytes(Dg.rng(0x7A05), 32).tobytes()     # same needle on every rank
        k, costs = 16, LEV
        if wl == "cfg5w":
            cs = args.costs.split(",")
            costs = (int(cs[0]), int(cs[1]), int(cs[2]), None if cs[3] in ("-", "None", "") else int(cs[3]))
        cells_unit, bytes_unit = 32, 1                              # per haystack byte: 32 cells, 1 byte read
        desc = "levenshtein_search 32B needle over a %d MiB random haystack shard per GPU, k=16, Best" % mib
        if wl == "cfg5w":
            desc += ", EditCosts(%d,%d,%d,%s)" % (costs[0], costs[1], costs[2], "None" if costs[3] is None else "Some(%d)" % costs[3])
        unit_name, dtype = "haystack bytes", "u16+u16 (cost|length packed in a u32 lane)"

        def make(share_of_common_batch):
            size = mib << 20
            if share_of_common_batch:                               # strong: one `mib` haystack cut into `world` shards
                lo, hi = TD.shard_range(size, rank, world)
                size = hi - lo
            g = Dg.rng(0x7A05 + 1000 * rank)
            hay_np = Dg.random_bytes(g, size)
            for pos in range(1 << 16, hay_np.size - 100, 1 << 20):     # ~1 planted mutated copy per MiB
                mm = np.frombuffer(Dg.mutate(g, needle, 10 if wl == "cfg5" else 5, wl == "cfg5w" and costs[3] is not None), dtype=np.uint8)
                hay_np[pos:pos + mm.size] = mm
            hay = B.haystack_tensor(hay_np)                         # resident in HBM from here on
            holder = {}
            if not dist_on:
                def run():
                    hits = B.levenshtein_search_best_dev(needle, hay, k, costs)        # kernels + on-device selection of the best-k hits
                    holder["best"] = TD.fold_best(hits, k, True)                       # the sequential Best pass (host)
            else:
                def run():                                          # the ranks' shards are one haystack: halo exchange + gather
                    holder["best"] = TD.levenshtein_search_sharded(needle, hay, k, T.SearchType.Best, T.EditCosts(*costs))

            def parity():
                run(); torch.cuda.synchronize()
                ns = min(hay_np.size, 4 << 20)
                want = O.levenshtein_search_naive_with_opts(needle, hay_np[:ns].tobytes(), k, O.ALL, costs, False)
                allhits = B.levenshtein_search_dev(needle, hay, k, costs)              # All-mode hits of this rank's shard
                got = [tuple(int(v) for v in r) for r in allhits if r[1] <= ns]
                assert got == [w for w in want if w[1] > 0], "parity gate failed: HIP search != oracle"
                extra_t["parity_some"] = len(got)                   # matches compared record for record
                if world == 1:
                    want_best = TD.fold_best(allhits, k, True)
                    got_best = [tuple(int(v) for v in m) for m in holder["best"]] if dist_on else holder["best"]
                    assert got_best == want_best, "parity gate failed: Best matches != fold over all hits"
                if dist_on:          # every rank must hold the same answer (the sharded search itself is compared with the monolithic
                    # oracle across shard cuts in tests/test_gpu_dist.py and tests/test_dist_cpu.py)
                    mine = torch.tensor([hash(tuple(tuple(int(v) for v in m) for m in holder["best"])) & 0x7FFFFFFF], dtype=torch.int64)
                    mine = mine.cuda() if backend == "nccl" else mine
                    allv = [torch.zeros_like(mine) for _ in range(world)]
                    dist.all_gather(allv, mine)
                    assert len({int(v[0].item()) for v in allv}) == 1, "sharded search: ranks disagree"
                return ns
            def end_to_end(reps=3):
                """host haystack in, Best matches out: pinned H2D of the shard + the pass (its report comes back by itself); ms (median)"""
                pin = torch.from_numpy(hay_np).pin_memory()
                ts = []
                for _ in range(reps + 1):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    hay[0][: pin.numel()].copy_(pin, non_blocking=True)
                    run()
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1) * 1e3)
                return float(np.median(ts[1:]))
            extra_t = {"hay_np": hay_np, "cells_total": cells_unit * hay_np.size, "bytes_total": bytes_unit * hay_np.size, "end_to_end": end_to_end}
            return run, hay_np.size, parity, extra_t

    # ------------------------------------------------------------------ timing helpers
    def barrier():
        if dist_on:
            dist.barrier()
        # wait for the device by polling an event (no sleep / wake-up of the host thread inside the timed region: a 20-step region
        # is 8 ms long), then the synchronize the protocol asks for -- which has nothing left to wait for
        done = torch.cuda.Event()
        done.record()
        while not done.query():
            pass
        torch.cuda.synchronize()

    def timed_region(run, steps, warmup):
        """clock ramp (untimed, --prewarm-ms), W untimed warm-ups, then EXACTLY `steps` passes between barrier + synchronize;
        max over ranks.  -> (wall seconds, mean device ms per pass from HIP events on the launch stream, ramp passes run)"""
        # The K timed passes as ONE hipGraph launch where a pass is nothing but kernel launches on the current stream (no scratch, no
        # host round trip): captured FIRST (the capture leaves the device idle for milliseconds: the clock ramp and the warm-ups come after
        # it), replayed inside the timed region.  Anything else -- or a failed capture -- enqueues the K passes one by one.
        graph = None
        if graphable and not args.no_graph:
            try:
                run()                                         # (one pass outside the capture: the library sizes its scratch lists here, not inside it)
                torch.cuda.synchronize()
                side = torch.cuda.Stream()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(steps):
                        run()
                torch.cuda.synchronize()
                graph.replay()                                # (once untimed: the first replay uploads the graph)
                torch.cuda.synchronize()
            except Exception as e:
                print("bench.py: hipGraph capture failed (%s: %s): K separate launches" % (type(e).__name__, e), file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
        n_ramp = 0
        if args.prewarm_ms > 0:
            # the number of ramp passes is agreed between the ranks (a pass may hold collectives): one timed pass, the maximum over ranks
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            n_ramp = int(min(4000, max(1, args.prewarm_ms / 1e3 / max(time.perf_counter() - t1, 1e-6))))
            if dist_on:
                tn = torch.tensor([n_ramp], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(tn, op=dist.ReduceOp.MAX)
                n_ramp = int(tn.item())
            for _ in range(n_ramp):
                run()
            n_ramp += 1                                       # (the pass that sized the ramp)
            torch.cuda.synchronize()
        for _ in range(warmup):
            run()
        launch_mode["mode"] = "one hipGraph of %d passes" % steps if graph is not None else "%d separate launches" % steps
        barrier()
        # two events around the region, on the stream the kernels are launched on (an event after every pass puts a barrier packet
        # between consecutive kernels: 0.6 - 8.7 us of idle device per pass, depending on the box)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        if graph is not None:
            graph.replay()
        else:
            for i in range(steps):
                run()
        ev1.record()
        barrier()
        elapsed = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1) / steps
        if dist_on:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, dev_ms, n_ramp

    def totals(*vals):
        """sums over the ranks"""
        if not dist_on:
            return [int(v) for v in vals]
        tt = torch.tensor([int(v) for v in vals], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt)
        return [int(v) for v in tt.tolist()]

    # ------------------------------------------------------------------ parity gate, warm-up, timed region
    # a pass that is a pure kernel launch on the current stream can be captured into a hipGraph: the fixed-length pair batches
    graphable = (not dist_on and wl in ("cfg1", "cfg2", "cfg4", "cfg2w", "cfg4w", "cfg2l", "cfg2s") and args.dist in ("random", "mutated")
                 and not args.early_out)
    launch_mode = {}
    run, units, parity, extra = make(strong and world > 1)
    parity_n = parity()
    parity_some, parity_side = extra.get("parity_some"), extra.get("parity_side", 0)
    if wl in ("cfg2t", "cfg2tp"):      # algorithmic bytes of a traceback pass: the strings, the distance and run count per pair, the runs written (16 / 4 B each)
        extra["bytes_total"] += 4 * units + (16 if wl == "cfg2t" else 4) * extra["runs_total"]
    info = T.last_launch_info()
    kernel_name = T.last_kernel_name()
    elapsed, dev_ms, n_ramp = timed_region(run, args.steps, args.warmup)
    # A second, DISCLOSED figure (never `value`): the same K passes with consecutive passes on TWO streams (one graph, fork / join inside the
    # capture, two output buffers).  A pass costs whole wavefronts per SIMD (15.26 per SIMD cost 16) and its first resident set waits for its
    # first lines together (~17 us): neither can be recovered INSIDE a pass, but a caller with batches in flight on two
    # streams lets the next pass's head fill this pass's tail.  Measured in the same process right after the timed region.
    overlapped = None
    if graphable and not args.no_graph and world == 1 and extra.get("run_alt") and args.steps >= 2:
        try:
            run_alt = extra["run_alt"]
            run_alt(); torch.cuda.synchronize()
            side, s2 = torch.cuda.Stream(), torch.cuda.Stream()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=side):
                s2.wait_stream(side)
                for i in range(args.steps):
                    if i & 1:
                        with torch.cuda.stream(s2):
                            run_alt()
                    else:
                        run()
                side.wait_stream(s2)
            torch.cuda.synchronize()
            g2.replay(); torch.cuda.synchronize()
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.steps
                best = ms if best is None or ms < best else best
            oa, ob = extra["out_pair"]
            assert torch.equal(oa, ob), "overlapped passes: the two streams' answers differ"
            overlapped = {"device_ms_per_pass": best, "value": extra["cells_total"] / (best / 1e3) / 1e9, "unit": "GCUPS",
                          "note": "NOT the headline: the same %d passes in one hipGraph with consecutive passes on two streams (they may overlap: the next "
                                  "pass's first wavefronts fill the tail of this one); best of 3 replays, HIP events" % args.steps}
        except Exception as e:
            print("bench.py: overlapped-passes figure failed (%s: %s)" % (type(e).__name__, e), file=sys.stderr)
            torch.cuda.synchronize()
    all_units, all_cells = totals(units, extra["cells_total"])
    e2e_ms = extra["end_to_end"]() if world == 1 else None

    strong_fig = None
    extra_bytes = extra["bytes_total"]
    if world > 1 and not strong:        # the second figure of a multi-rank weak run: the same batch partitioned over the ranks
        del run, parity, extra
        torch.cuda.empty_cache()
        run_s, units_s, parity_s, extra_s = make(True)
        parity_s()
        el_s, dev_s_ms, _ = timed_region(run_s, args.steps, args.warmup)
        tot_s, cells_s = totals(units_s, extra_s["cells_total"])
        if rank == 0:
            strong_fig = {"value": cells_s * args.steps / el_s / 1e9, "unit": "GCUPS", "ms_per_step": el_s / args.steps * 1e3,
                          "units_total": tot_s, "units_this_rank": units_s, "device_ms_per_pass": dev_s_ms}
            ppw = (T.last_launch_info() or {}).get("pairs_per_wave") or 0
            if ppw and wl not in ("cfg5", "cfg5w", "hsearch"):
                # why strong scaling of a sub-millisecond pass flattens: a pass costs WHOLE wavefronts per SIMD, and a
                # rank's share below one resident set (256 CUs x 16 wavefronts) still costs one wavefront lifetime + the pass's fixed part
                wf = -(-units_s // ppw)
                strong_fig["resident_set_arithmetic"] = {
                    "pairs_per_wavefront": ppw, "wavefronts_this_rank": wf, "resident_set_wavefronts": 4096,
                    "resident_sets_this_rank": round(wf / 4096.0, 3),
                    "note": "below ~1 resident set per rank a pass costs one wavefront lifetime + its fixed part, not 1/N of the 1-GPU pass: "
                            "weak scaling (the line's value) is the curve that stays flat; this figure is expected to flatten beyond N ~ %d"
                            % max(1, (units // ppw) // 4096)}
        extra = {"bytes_total": extra_bytes}
    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    value = all_cells * args.steps / elapsed / 1e9
    dev_s = dev_ms / 1e3                                   # device time of one pass (HIP events on the launch stream)
    achieved = extra["bytes_total"] / dev_s / 1e9

    # committed counter passes of the same command (profiles/<round>/): HBM-side traffic and the VALU-issue roofline
    def load_json(*parts):
        p = os.path.join(ROOT, "profiles", *parts)
        try:
            return json.load(open(p))
        except Exception:
            return None
    traffic = None
    traffic_source = None
    valu_issue = None

    def issue_roofline(c, source):
        """The bound that matters on this integer path: cycles per wave64 VALU instruction per SIMD over the dominant kernel's launch
        (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs), against the guide's 2-cycle ceiling and against what a MIXED stream of
        full-rate and half-rate opcodes issues at on gfx950 (scripts/ubench_mix.hip, measured; committed under profiles/)."""
        insts, busy = c["SQ_INSTS_VALU"]["mean_per_launch"], c["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
        cyc_per_inst = 1024.0 * busy / insts
        v = {"kernel": c.get("_dominant"), "valu_insts_per_launch": insts, "busy_cycles_per_xcd": busy, "simds": 1024,
             "cycles_per_valu_inst_per_simd": cyc_per_inst,
             # MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32 -> the hard ceiling
             "frac_of_2cycle_ceiling": 2.0 / cyc_per_inst, "source": source}
        try:
            mix = load_json(PROFILE_ROUND, "isa_mix.json")
            if mix and tag in mix:          # opcode histogram of the inner loop x per-class issue cost measured opcode by opcode
                m = mix[tag]
                v["isa_model_cycles_per_valu_inst"] = m["modelled_cycles_per_valu_inst"]
                v["frac_of_isa_model"] = m["modelled_cycles_per_valu_inst"] / cyc_per_inst
            ub = open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "ubench_mix.txt")).read()
            sect = ub[ub.index("4 wave(s) per SIMD"):]
            mixed = float(sect[sect.index("xor/perm strictly alternating"):].split("ms")[1].split("cycles")[0])
            v["measured_mixed_stream_cycles_per_valu_inst"] = mixed
            v["frac_of_mixed_stream_rate"] = mixed / cyc_per_inst
        except Exception:
            pass
        return v
    tag = wl + ("" if args.dist in ("random", "mutated") else "_" + args.dist) + ("%d" % args.needle_len if wl == "hsearch" else "") + \
          ("_" + args.costs.replace(",", "") if wl == "cfg5w" else "")       # the profile files of this workload / distribution
    pmc = load_json(PROFILE_ROUND, "bench_%s_pmc.json" % tag)
    # a committed counter pass is only spliced into the line when it was recorded for the kernel this run launched
    # (T.last_kernel_name(): the dominant kernel of the pass, as rocprofv3 prints it) -- never for another build's kernel
    if pmc and kernel_name and kernel_name not in str(pmc.get("_dominant", "")):
        print("bench.py: profiles/%s counter pass was recorded for %r, this run launched %r: not spliced"
              % (PROFILE_ROUND, pmc.get("_dominant"), kernel_name), file=sys.stderr)
        pmc = None
    # measured in THIS run where rocprofv3 is on the box: scripts/pmc_collect.py (separate --pmc passes around this same command, 3 steps)
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "RPD_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if not args.no_pmc and world == 1 and not dist_on and not under_profiler:       # (never a profiler inside a profiler)
        import shutil
        import tempfile
        if shutil.which("rocprofv3"):
            try:
                tmp_json = os.path.join(tempfile.mkdtemp(prefix="ta_pmc_"), "pmc.json")
                flags = ["--dist", args.dist] + (["--pairs", str(args.pairs)] if args.pairs else []) + \
                        (["--needle-len", str(args.needle_len)] if wl == "hsearch" else []) + (["--costs", args.costs] if wl == "cfg5w" else []) + (["--early-out"] if args.early_out else []) + (["--unit-prefilter"] if args.unit_prefilter else []) + \
                        ["--prewarm-ms", "0"]
                r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_collect.py"), "--out", tmp_json, "--workload", wl,
                                    "--sets", "rd_b,write,issue", "--steps", "3", "--extra", " ".join(flags)],
                                   capture_output=True, text=True, timeout=360)
                live = json.load(open(tmp_json))
                if kernel_name and kernel_name in str(live.get("_dominant", "")):
                    traffic = int(live["_traffic"]["bytes_per_pass"])
                    traffic_source = ("measured in this run: rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_64B/128B, WRITE_SIZE (separate passes, "
                                      "3 steps each, same workload) after the timed region; 128 x RDREQ_128B + 64 x RDREQ_64B + WRITE_SIZE, "
                                      "all kernels of one pass")
                    if "SQ_INSTS_VALU" in live and "GRBM_GUI_ACTIVE" in live:
                        valu_issue = issue_roofline(live, "measured in this run: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE "
                                                          "(its own pass, 3 steps, same workload) after the timed region")
            except Exception as e:
                print("bench.py: live counter pass failed (%s: %s): replaying the committed one" % (type(e).__name__, e), file=sys.stderr)
    if pmc and traffic is None:
        try:
            traffic = int(pmc["_traffic"]["bytes_per_pass"])      # size-resolved L2 fabric-side requests, all kernels of one pass
            traffic_source = ("replayed from the committed counter pass profiles/%s/bench_%s_pmc.json (rocprofv3 --pmc, separate passes, "
                              "same command, kernel %s); not measured in this run" % (PROFILE_ROUND, tag, pmc.get("_dominant")))
        except Exception:
            traffic = None
    if pmc and valu_issue is None:
        try:
            valu_issue = issue_roofline(pmc, "replayed from profiles/%s/bench_%s_pmc.json; not measured in this run" % (PROFILE_ROUND, tag))
        except Exception:
            pass

    cpu = None
    if not args.no_cpu and world == 1:        # the CPU leg runs at N = 1 only (rank 0)
        facts = host_cpu_facts()

        def timed(fn, min_s=3.0, max_reps=64):
            """fn() repeatedly until min_s has passed -> (seconds per call, calls)"""
            fn()                                                        # page in, spin the OpenMP team up
            t, reps = time.perf_counter(), 0
            while True:
                fn(); reps += 1
                dt = time.perf_counter() - t
                if dt >= min_s or reps >= max_reps:
                    return dt / reps, reps
        if wl == "hsearch":
            hay_np = extra["hay_np"]
            cpu_sample = min(64 << 20, hay_np.size)
            t1 = time.perf_counter()
            O.hamming_search_naive_with_opts(needle, hay_np[:cpu_sample].tobytes(), k, O.ALL)
            dt = time.perf_counter() - t1
            cpu = {"value": cells_unit * cpu_sample / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
                   "sample": "first %d MiB of the shard, single thread, oracle/ta_oracle.c (restated scalar hamming_search_naive), %.1f s"
                             % (cpu_sample >> 20, dt), "host": facts}
        elif wl in ("cfg5", "cfg5w"):
            hay_np = extra["hay_np"]
            cpu_sample = min(8 << 20, hay_np.size)
            t1 = time.perf_counter()
            O.levenshtein_search_naive_with_opts(needle, hay_np[:cpu_sample].tobytes(), k, O.BEST, costs, False)
            dt = time.perf_counter() - t1
            cpu = {"value": cells_unit * cpu_sample / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
                   "sample": "first %d MiB of the shard, single thread (the scalar search is one serial scan), oracle/ta_oracle.c "
                             "(restated scalar path), %.1f s" % (cpu_sample >> 20, dt), "host": facts}
        elif wl in ("cfg2t", "cfg2tp"):
            a_np, b_np = extra["csr"](0, min(units, 2000))
            sample = min(units, 2000)
            sa_l = [bytes(a_np[0][int(a_np[1][i]):int(a_np[1][i + 1])]) for i in range(sample)]
            sb_l = [bytes(b_np[0][int(b_np[1][i]):int(b_np[1][i + 1])]) for i in range(sample)]
            t1 = time.perf_counter()
            for x, y in zip(sa_l, sb_l):
                O.levenshtein_simd_k_with_opts(x, y, k, True, costs)
            dt = time.perf_counter() - t1
            cpu = {"value": cells_unit * sample / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
                   "sample": "first %d pairs, one thread, oracle/ta_oracle.c (restated scalar banded path with its traceback), %.2f s" % (sample, dt), "host": facts}
        elif wl in ("cfg2", "cfg4", "cfg2w", "cfg4w", "cfg2l", "cfg2s"):
            # Inputs staged once (CSR blobs), outside the timed loops.  Three restatements: the hand-written AVX2 one with
            # saturating u8 cells (oracle/ta_oracle_avx2.c: 64 / 32 u8 lanes per anti-diagonal for cfg2 / cfg4 -- the reference's
            # own Avx2x32x8 / Avx1x32x8 classes), the compiler-vectorised u16 anti-diagonal one (ta_oracle_simd.c) and the scalar
            # one (ta_oracle.c).  `value` is the best all-thread figure; `cores` the EFFECTIVE parallelism it reached
            # (all-thread rate / one-thread rate), with the thread-scaling line that shows where it saturates.
            ns = min(units, 1_000_000)
            ca, cb = extra["csr"](0, ns)
            n1 = min(ns, 20_000)
            c1a, c1b = extra["csr"](0, n1)
            ref = O.levenshtein_k_batch(c1a, c1b, k, costs, threads=cores)
            variants = [("scalar u32 (oracle/ta_oracle.c)", O.levenshtein_k_batch)]
            got = O.levenshtein_k_batch_antidiag(c1a, c1b, k, costs, threads=cores)       # (None: affine gaps are not restated there)
            if got is not None:
                assert np.array_equal(got, ref), "the CPU restatements differ (u16 anti-diagonal)"
                variants.append(("anti-diagonal u16, compiler-vectorised (oracle/ta_oracle_simd.c)", O.levenshtein_k_batch_antidiag))
            have_u8 = O.have_avx2()
            if have_u8:
                got, hist = O.levenshtein_k_batch_ladder(c1a, c1b, k, costs, threads=cores, hist=True)
                if got is not None:
                    assert np.array_equal(got, ref), "the CPU restatements differ (AVX2 u8 anti-diagonal)"
                    lanes = [0, 32, 64, 128, 256, -1][int(np.argmax(hist))]
                    variants.append(("anti-diagonal AVX2, %d saturating u8 lanes (oracle/ta_oracle_avx2.c)" % lanes, O.levenshtein_k_batch_ladder))
            gcups = lambda nu, sec: extra["cells_total"] / units * nu / sec / 1e9
            res = {}
            for name, fn in variants:
                s_all, _ = timed(lambda: fn(ca, cb, k, costs, threads=cores), min_s=2.5)
                s_one, _ = timed(lambda: fn(c1a, c1b, k, costs, threads=1), min_s=1.0, max_reps=4)
                res[name] = (gcups(ns, s_all), gcups(n1, s_one))
            best = max(res, key=lambda nm: res[nm][0])
            fn_best = dict(variants)[best]
            scaling_line = []
            th = 1
            while th < cores:
                s_t, _ = timed(lambda: fn_best(ca, cb, k, costs, threads=th), min_s=0.8, max_reps=3)
                scaling_line.append((th, round(gcups(ns, s_t), 2)))
                th *= 2
            scaling_line.append((cores, round(res[best][0], 2)))
            eff = res[best][0] / res[best][1]
            cpu = {"value": res[best][0], "unit": "GCUPS", "cores": round(eff, 1), "kind": "port",
                   "sample": "%d pairs of the same batch staged once; best variant: %s, %.1f GCUPS on %d OpenMP threads = %.1fx its "
                             "one-thread rate of %.2f GCUPS (effective cores; the host offers nproc %s, affinity %s, cgroup quota %s); "
                             "each variant repeated for >= 2.5 s"
                             % (ns, best, res[best][0], cores, eff, res[best][1], facts["nproc"], facts["affinity"], facts["cgroup_cpus"]),
                   "threads_used": cores, "host": facts,
                   "variants_gcups_all_threads_and_one_thread": {nm: [round(v[0], 2), round(v[1], 3)] for nm, v in res.items()},
                   "thread_scaling_gcups": scaling_line}
        else:
            oracle = extra["oracle"]
            cpu_sample = units if wl == "cfg1" else max(cores, 64)
            t1 = time.perf_counter()
            oracle(0, cpu_sample, cores)
            dt = time.perf_counter() - t1
            t2 = time.perf_counter()
            n_one = max(1, cpu_sample // max(cores, 1)) if wl == "cfg3" else cpu_sample
            oracle(0, n_one, 1)
            dt1 = time.perf_counter() - t2
            v_all, v_one = cells_unit * cpu_sample / dt / 1e9, cells_unit * n_one / dt1 / 1e9
            cpu = {"value": v_all, "unit": "GCUPS", "cores": round(v_all / v_one, 1), "kind": "port",
                   "sample": "first %d %s of the same batch, %d OpenMP threads, oracle/ta_oracle.c (restated scalar path), %.1f s; one "
                             "thread: %.3f GCUPS (cores = all-thread / one-thread rate)" % (cpu_sample, unit_name, cores, dt, v_one),
                   "threads_used": cores, "host": facts}

    if info.get("kernel") == 3:
        dtype = "u32 bit-vectors, 1 bit per band cell (reference width class u%d)" % info.get("cell_bits", 8)
    if args.dist == "ragged":
        evaluated_unit = None                              # per-pair geometry: only the credited figure is reported
    evaluated_value = value * evaluated_unit / cells_unit if evaluated_unit else None
    line = {
        "metric": "GCUPS (DP cell updates/s) for k-banded Levenshtein, 1M x 256B pairs" if wl == "cfg2" else "GCUPS (%s)" % wl,
        "value": value, "unit": "GCUPS",
        "value_is": ("weak scaling: every GPU its own batch of the configured size (%d units in all); the SAME batch split over the GPUs is `strong_scaling`" % all_units)
                    if (world > 1 and not strong) else ("strong scaling: ONE batch of the configured size split over the GPUs" if world > 1 else "one GPU"),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_ms": args.prewarm_ms,
        "prewarm_passes": n_ramp,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": "%s: %s (%s bytes)" % (wl, desc, args.dist), "units_per_gpu": units, "units_total": all_units,
                   "unit": unit_name, "credited_cells_per_unit": all_cells / max(all_units, 1), "evaluated_band_cells_per_unit": evaluated_unit,
                   "parallelism": "independent units sharded x%d (%s), %s" % (
                       world, args.scaling, "no collective" if wl not in ("cfg5", "cfg5w") or not dist_on else
                       "halo tails + match lists all-gathered (%s)" % ("RCCL" if backend == "nccl" else backend)),
                   "backend": backend if dist_on else None,
                   "early_out": bool(args.early_out), "unit_prefilter": bool(args.unit_prefilter)},
        "value_evaluated_cells": evaluated_value,
        "end_to_end_ms": e2e_ms,          # host buffers in, answers out (pinned H2D + pass + D2H); never the headline
        "strong_scaling": strong_fig,
        "overlapped_passes": overlapped,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel_name": kernel_name,
                     "device_ms_per_pass": dev_s * 1e3, "algorithmic_bytes_per_pass": extra["bytes_total"],
                     "valu_issue": valu_issue,
                     "note": "integer VALU-issue-bound path (DESIGN.md section 5); the HBM fraction is reported because north_star asks for it"},
        "cpu_baseline": cpu,
        "timed_region": launch_mode.get("mode"),
        "kernel": info, "parity_checked_units": parity_n + parity_side,
        # how many of the answers compared with the oracle before the timed region were Some(d) -- random pairs at these k are all None,
        # so the gate also runs a side batch of mutated pairs of the same geometry through the same entry point (parity_side_batch)
        "parity_checked_some": parity_some, "parity_side_batch": parity_side,
    }
    # ------------------------------------------------------------------ BASELINE.json's other configs, in the same driver run
    # (four of five configs were builder-run claims.)  The default invocation keeps cfg2 as the headline -- everything above --
    # and then times cfg4, cfg5 (1 GiB), cfg3 and cfg1 each in its own bench.py process: its own parity gate, fewer steps, its own live
    # counter passes.  Their figures ride in `all_configs`; a leg that fails reports its error and the line is printed regardless.
    if wl == "cfg2" and world == 1 and not dist_on and args.dist == "random" and not args.no_all_configs and not args.no_pmc and \
            not args.early_out and not args.unit_prefilter and not args.pairs and not under_profiler:
        rows = [{"workload": line["config"]["workload"], "ms_per_step": line["ms_per_step"], "value": value, "unit": "GCUPS",
                 "roofline_frac": achieved / HBM_PEAK_GBS, "cycles_per_valu_inst": (valu_issue or {}).get("cycles_per_valu_inst_per_simd"),
                 "traffic_ratio": (traffic / extra["bytes_total"]) if traffic else None, "kernel_name": kernel_name, "steps": args.steps,
                 "parity_checked_some": parity_some}]
        legs = [("cfg4", 20), ("cfg5", 10), ("cfg3", 3), ("cfg1", 100)]
        t_legs = time.perf_counter()
        for leg, leg_steps in legs:
            row = {"workload": leg}
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", leg, "--steps", str(leg_steps), "--warmup", "2",
                                    "--no-cpu", "--no-all-configs", "--prewarm-ms", "150"], capture_output=True, text=True, timeout=420)
                js = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode != 0 or not js:
                    raise RuntimeError("rc %d: %s" % (r.returncode, r.stderr[-300:]))
                d = json.loads(js[-1])
                rf = d["roofline"]
                row = {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"],
                       "roofline_frac": rf["frac"], "cycles_per_valu_inst": (rf.get("valu_issue") or {}).get("cycles_per_valu_inst_per_simd"),
                       "traffic_ratio": (rf["traffic"] / rf["algorithmic_bytes_per_pass"]) if rf.get("traffic") else None,
                       "kernel_name": rf.get("kernel_name"), "steps": d["steps"], "device_ms_per_pass": rf.get("device_ms_per_pass"),
                       "parity_checked_some": d.get("parity_checked_some"), "end_to_end_ms": d.get("end_to_end_ms")}
                if leg == "cfg1":
                    row["note"] = ("20 MB batch re-read every pass: resident in the 256 MB Infinity Cache and launch-bound -- roofline_frac is a "
                                   "cache-side figure here, not an HBM one")
            except Exception as e:
                row["error"] = "%s: %s" % (type(e).__name__, e)
            rows.append(row)
        line["all_configs"] = rows
        line["all_configs_seconds"] = round(time.perf_counter() - t_legs, 1)
    print(json.dumps(line))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
