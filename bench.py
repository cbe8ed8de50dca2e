#!/usr/bin/env python3
"""bench.py -- GCUPS of the edit-distance hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM.
Default workload = BASELINE.json configs[1] (cfg2): levenshtein_simd_k, k = 32, 1M random 256-byte pairs,
LEVENSHTEIN_COSTS -- the configuration the metric is quoted on.  The other configs are parity-test cases;
they can be timed with --workload for DESIGN.md but are not the bench line.

With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank processes its own batch
(independent units: weak scaling, no data-path collective); rank 0 prints ONE JSON line carrying
`roofline` (HBM: algorithmic bytes / measured kernel time) and `cpu_baseline` (the CPU oracle on a bounded
sample, all host cores: the anti-diagonal vectorised restatement for cfg2/cfg4 with the scalar figure beside
it, the scalar restatement elsewhere).
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

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--pairs", type=int, default=0, help="override the number of pairs (cfg5: haystack MiB) per GPU")
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
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU fallback)"
    local = local % torch.cuda.device_count()          # (lets a 1-GPU box dry-run the multi-rank control flow)
    torch.cuda.set_device(local)
    dist = None
    backend = os.environ.get("TA_BENCH_BACKEND", "nccl")   # "nccl" == RCCL on ROCm; "gloo" only for dry runs
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    wl = args.workload
    evaluated_unit = None
    seed = 0x7A00 + int(wl[3:]) + 1000 * rank
    cores = O.max_threads()
    LEV, RDAM = (1, 1, 0, None), (1, 1, 0, 1)

    # ------------------------------------------------------------------ workload set-up
    if wl in ("cfg2", "cfg4", "cfg3", "cfg1"):
        n, L, k, costs = {"cfg1": (10_000, 1024, None, None), "cfg2": (1_000_000, 256, 32, LEV),
                          "cfg3": (100_000, 4096, None, LEV), "cfg4": (1_000_000, 128, 8, RDAM)}[wl]
        n = args.pairs or n
        if args.dist == "random":
            a, b = Dg.pairs_random(seed, n, L)
        else:
            g = Dg.rng(seed)
            a = g.integers(33, 127, size=(n, L), dtype=np.uint8)
            b = a.copy()
            kk = k or 64
            pos = g.integers(0, L, size=(n, max(1, kk // 2)))
            b[np.arange(n)[:, None], pos] = 32
        sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        bytes_unit = 2 * L + 4
        if wl == "cfg1":
            cells_unit = L
            run = lambda: B.hamming_batch(sa, sb, out=out)
            oracle = lambda lo, hi, th: O.hamming_batch(O.csr_from_fixed(a[lo:hi]), O.csr_from_fixed(b[lo:hi]), threads=th)
            desc, unit_name, dtype = "hamming() on 10K random 1KiB pairs (GPU batch kernel)", "byte pairs", "u8"
            cpu_sample = n
        elif wl == "cfg3":
            cells_unit = L * L                                     # the answer's work: the full matrix
            run = lambda: B.levenshtein_exp_batch(sa, sb, costs, out=out)
            oracle = lambda lo, hi, th: O.levenshtein_exp_batch(O.csr_from_fixed(a[lo:hi]), O.csr_from_fixed(b[lo:hi]), costs, threads=th)
            desc, unit_name, dtype = "levenshtein_exp full distance on 100K random 4KiB pairs", "pairs", "u32"
            cpu_sample = max(cores, 64)
        else:
            cells_unit = O.band_cells(L, L, k, costs)              # cells the scalar banded path visits
            run = lambda: B.levenshtein_k_batch(sa, sb, k, costs, out=out)
            oracle = lambda lo, hi, th: O.levenshtein_k_batch(O.csr_from_fixed(a[lo:hi]), O.csr_from_fixed(b[lo:hi]), k, costs, threads=th)
            desc = {"cfg2": "levenshtein_simd_k k=32, 1M random 256B pairs, u8 cells",
                    "cfg4": "levenshtein_simd_k_with_opts RDAMERAU_COSTS k=8, 1M 128B pairs (transposition path)"}[wl]
            unit_name, dtype = "pairs", "u32"   # reference width class u8 (ta_levenshtein_select), arithmetic in 32-bit VGPR lanes
            cpu_sample = min(n, 20000 * max(1, cores // 2))
            # cells inside the band the kernels evaluate: [min(0,delta) - t, max(0,delta) + t], t = (unit_k - |delta|) / 2
            #  -- about half of the credited reference band; reported for transparency, never credited
            uk = min((min(k, L * max(costs[0], costs[1])) - costs[2]) // costs[1], 2 * L)
            evaluated_unit = sum(min(L, i + uk // 2) - max(1, i - uk // 2) + 1 for i in range(1, L + 1))
        units = n

        def parity():
            run(); torch.cuda.synchronize()
            ns = min(n, 4000 if wl != "cfg3" else 48)
            got = out[:ns].cpu().numpy().view(np.uint32)
            assert np.array_equal(got, oracle(0, ns, cores)), "parity gate failed: HIP path != oracle"
            return ns
    else:   # cfg5: levenshtein_search, 32 B needle over a 1 GiB random shard per GPU
        mib = args.pairs or 1024
        g = Dg.rng(seed)
        needle = Dg.random_bytes(Dg.rng(0x7A05), 32).tobytes()     # same needle on every rank
        hay_np = Dg.random_bytes(g, mib << 20)
        for pos in range(1 << 16, hay_np.size - 100, 1 << 20):     # ~1 planted mutated copy per MiB
            mm = np.frombuffer(Dg.mutate(g, needle, 10), dtype=np.uint8)
            hay_np[pos:pos + mm.size] = mm
        hay = B.haystack_tensor(hay_np)
        k, costs = 16, LEV
        cells_unit, bytes_unit, units = 32, 1, hay_np.size         # per haystack byte: 32 cells, 1 byte read
        holder = {}
        from triple_accel_amd import dist as TD

        def run():
            hits = B.levenshtein_search_best_dev(needle, hay, k, costs)            # kernels + on-device selection of the best-k hits
            holder["best"] = TD.fold_best(hits, k, True)                           # the sequential Best pass (host)
        desc = "levenshtein_search 32B needle over a %d MiB random haystack shard per GPU, k=16, Best" % mib
        unit_name, dtype = "haystack bytes", "u16+u16 (cost|length packed in a u32 lane)"
        cpu_sample = 8 << 20

        def oracle_search(lo, hi, th):
            return O.levenshtein_search_naive_with_opts(needle, hay_np[lo:hi].tobytes(), k, O.BEST, costs, False)

        def parity():
            run(); torch.cuda.synchronize()
            ns = min(hay_np.size, 4 << 20)
            want = O.levenshtein_search_naive_with_opts(needle, hay_np[:ns].tobytes(), k, O.ALL, costs, False)
            allhits = B.levenshtein_search_dev(needle, hay, k, costs)              # All-mode hits of the whole shard
            got = [tuple(int(v) for v in r) for r in allhits if r[1] <= ns]
            assert got == [w for w in want if w[1] > 0], "parity gate failed: HIP search != oracle"
            assert holder["best"] == TD.fold_best(allhits, k, True), "parity gate failed: on-device Best selection != fold over all hits"
            return ns

    # ------------------------------------------------------------------ parity gate, warm-up, timed region
    parity_n = parity()
    info = T.last_launch_info()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        run()
        ev[i + 1].record()                                # same stream as the kernel launches
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = cells_unit * units * args.steps * world / elapsed / 1e9
    dev_s = float(np.mean(step_ms)) / 1e3                 # device time of one pass (HIP events on the launch stream)
    achieved = bytes_unit * units / dev_s / 1e9
    traffic = None
    prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(prof):
        try:
            traffic = json.load(open(prof)).get(wl, {}).get("bytes_per_launch")
        except Exception:
            traffic = None

    # the companion roofline of this integer path: VALU issue (one instruction per SIMD per 4 cycles), from the committed
    # counter passes of the same command (profiles/r01/bench_cfg2_pmc.json); None when no profile is present
    valu_issue = None
    pmc = os.path.join(ROOT, "profiles", "r01", "bench_%s_pmc.json" % wl)
    if os.path.exists(pmc):
        try:
            c = json.load(open(pmc))
            insts, busy = c["SQ_INSTS_VALU"]["mean_per_launch"], c["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
            valu_issue = {"valu_insts_per_launch": insts, "busy_cycles_per_xcd": busy, "simds": 1024,
                          "frac": insts * 4.0 / (1024.0 * busy), "source": "profiles/r01/bench_%s_pmc.json" % wl}
        except Exception:
            valu_issue = None

    cpu = None
    if not args.no_cpu and world == 1:        # the CPU leg runs at N = 1 only (rank 0)
        def timed(fn, min_s=4.0, max_reps=64):
            """fn() repeatedly until min_s has passed -> (seconds per call, calls)"""
            fn()                                                        # page in, spin the OpenMP team up
            t, reps = time.perf_counter(), 0
            while True:
                fn(); reps += 1
                dt = time.perf_counter() - t
                if dt >= min_s or reps >= max_reps:
                    return dt / reps, reps
        if wl == "cfg5":
            t1 = time.perf_counter()
            oracle_search(0, cpu_sample, cores)
            dt = time.perf_counter() - t1
            cpu = {"value": cells_unit * cpu_sample / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
                   "sample": "first %d MiB of the shard, single thread (the scalar search is one serial scan), oracle/ta_oracle.c "
                             "(restated scalar path), %.1f s" % (cpu_sample >> 20, dt)}
        elif wl in ("cfg2", "cfg4"):
            # inputs staged once (CSR blobs), outside the timed loops; three figures: the scalar restatement on all host
            # threads, the anti-diagonal compiler-vectorised restatement (oracle/ta_oracle_simd.c: u16 cells, AVX2 when the
            # host has it -- the shape of the reference's own SIMD core) on all host threads and on one thread.  `value` is
            # the best all-thread figure.
            ns = min(n, 1_000_000)
            ca, cb = O.csr_from_fixed(a[:ns]), O.csr_from_fixed(b[:ns])
            n1 = min(ns, 20_000)
            c1a, c1b = O.csr_from_fixed(a[:n1]), O.csr_from_fixed(b[:n1])
            ref = O.levenshtein_k_batch(c1a, c1b, k, costs, threads=cores)
            got = O.levenshtein_k_batch_antidiag(c1a, c1b, k, costs, threads=cores)
            assert got is not None and np.array_equal(got, ref), "the two CPU restatements differ"
            s_sc, r_sc = timed(lambda: O.levenshtein_k_batch(ca, cb, k, costs, threads=cores))
            s_ad, r_ad = timed(lambda: O.levenshtein_k_batch_antidiag(ca, cb, k, costs, threads=cores))
            s_a1, r_a1 = timed(lambda: O.levenshtein_k_batch_antidiag(c1a, c1b, k, costs, threads=1), min_s=2.0)
            s_s1, r_s1 = timed(lambda: O.levenshtein_k_batch(c1a, c1b, k, costs, threads=1), min_s=2.0, max_reps=4)
            gcups = lambda units, sec: cells_unit * units / sec / 1e9
            v_sc, v_ad = gcups(ns, s_sc), gcups(ns, s_ad)
            cpu = {"value": max(v_sc, v_ad), "unit": "GCUPS", "cores": cores, "kind": "port",
                   "sample": "%d pairs of the same batch staged once, %d OpenMP threads, repeated for >= 4 s per variant "
                             "(%d / %d passes): anti-diagonal compiler-vectorised restatement (oracle/ta_oracle_simd.c) %.1f GCUPS, "
                             "scalar restatement (oracle/ta_oracle.c) %.1f GCUPS; one thread on %d pairs: %.2f / %.2f GCUPS"
                             % (ns, cores, r_ad, r_sc, v_ad, v_sc, n1, gcups(n1, s_a1), gcups(n1, s_s1)),
                   "antidiag_value": v_ad, "scalar_value": v_sc,
                   "antidiag_one_thread": gcups(n1, s_a1), "scalar_one_thread": gcups(n1, s_s1)}
        else:
            t1 = time.perf_counter()
            oracle(0, cpu_sample, cores)
            dt = time.perf_counter() - t1
            cpu = {"value": cells_unit * cpu_sample / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
                   "sample": "first %d %s of the same batch, %d OpenMP threads, oracle/ta_oracle.c (restated scalar path), %.1f s"
                             % (cpu_sample, unit_name, cores, dt)}

    if info.get("kernel") == 3:
        dtype = "u32 bit-vectors, 1 bit per band cell (reference width class u%d)" % info.get("cell_bits", 8)
    line = {
        "metric": "GCUPS (DP cell updates/s) for k-banded Levenshtein, 1M x 256B pairs" if wl == "cfg2" else "GCUPS (%s)" % wl,
        "value": value, "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": "%s: %s (%s bytes)" % (wl, desc, args.dist), "units_per_gpu": units, "unit": unit_name,
                   "credited_cells_per_unit": cells_unit, "evaluated_band_cells_per_unit": evaluated_unit, "parallelism": "independent units sharded x%d, no collective" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "device_ms_per_pass": dev_s * 1e3, "algorithmic_bytes_per_pass": bytes_unit * units,
                     "valu_issue": valu_issue,
                     "note": "integer VALU-issue-bound path (DESIGN.md section 5); the HBM fraction is reported because north_star asks for it"},
        "cpu_baseline": cpu,
        "kernel": info, "parity_checked_units": parity_n,
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
