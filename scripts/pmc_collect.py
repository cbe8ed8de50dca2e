#!/usr/bin/env python3
"""Run `rocprofv3 --kernel-trace --pmc <set>` once per counter set around a bench.py command (separate passes, as the pool's
gpurun requires) and aggregate every counter per kernel: mean per launch over the launches of the bench's timed steps.

    python scripts/pmc_collect.py --out profiles/r02/bench_cfg2_pmc.json --workload cfg2 [--sets sq,tcc] [--steps 5]

Output: {"<kernel>": {"launches": n, "<COUNTER>": {"mean_per_launch": x}}, ..., "_dominant": "<kernel with the most FETCH or
VALU>", plus the dominant kernel's counters flattened at the top level (what bench.py reads)}.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = {
    "sq1": "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY",
    "sq2": "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SMEM",
    "issue": "SQ_INSTS_VALU GRBM_GUI_ACTIVE",          # the VALU-issue roofline in ONE pass (bench.py's live counter leg)
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    "rd_a": "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum",
    "rd_b": "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum",
    "wr": "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum",
    "hit": "TCC_HIT_sum TCC_MISS_sum",
    "req": "TCC_REQ_sum TCC_READ_sum",
}
GROUPS = {"sq": ["sq1", "sq2"], "tcc": ["fetch", "write", "rd_a", "rd_b", "wr", "hit"], "fw": ["fetch", "write"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--sets", default="sq,tcc")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--extra", default="", help="extra bench.py flags")
    args = ap.parse_args()
    names = []
    for s in args.sets.split(","):
        names += GROUPS.get(s, [s])
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    tmp = "/tmp/pmc_collect_%d" % os.getpid()
    for nm in names:
        shutil.rmtree(tmp, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + SETS[nm].split() + ["-d", tmp, "-o", "p", "-f", "csv", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--steps", str(args.steps), "--warmup", "1", "--no-cpu", "--no-pmc", "--no-all-configs", "--no-side-batch"] + ([] if "--prewarm-ms" in args.extra else ["--prewarm-ms", "0"]) + args.extra.split()
        r = subprocess.run(cmd, cwd="/tmp", capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp"))
        files = glob.glob(tmp + "/**/p_counter_collection.csv", recursive=True)
        if r.returncode != 0 or not files:
            print("pass %s failed (rc %d): %s" % (nm, r.returncode, r.stderr[-400:]), file=sys.stderr)
            continue
        for row in csv.DictReader(open(files[0])):
            k = re.sub(r"\(.*", "", row["Kernel_Name"])
            if not k.startswith(("void ta::", "ta::")):
                continue
            per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for k, cs in per.items():
        # the timed steps' launches: drop the parity call and the warm-up when a kernel runs once per step (keep all otherwise)
        out[k] = {"launches": max(len(v) for v in cs.values())}
        for c, v in cs.items():
            vv = v[-args.steps:] if len(v) >= args.steps else v
            out[k][c] = {"mean_per_launch": sum(vv) / len(vv), "launches_averaged": len(vv), "launches_seen": len(v), "sum_all_launches": sum(v)}
    def weight(k):
        d = out[k]
        return d.get("SQ_INSTS_VALU", {}).get("mean_per_launch", 0) * d["launches"] + d.get("FETCH_SIZE", {}).get("mean_per_launch", 0) * d["launches"]
    if out:
        dom = max(out, key=weight)
        flat = dict(out[dom])
        flat["_dominant"] = dom
        flat["_kernels"] = out
        # fabric-side bytes of one bench pass, every kernel of the path included: size-resolved read requests (32 / 64 / 128 B) +
        # 64-byte write requests, summed over all launches and divided by the passes (= launches of the dominant kernel)
        passes = out[dom]["launches"]
        def tot(c):
            return sum(d[c]["sum_all_launches"] for d in out.values() if c in d)
        if any("TCC_EA0_RDREQ_128B_sum" in d for d in out.values()):
            rd = 128.0 * tot("TCC_EA0_RDREQ_128B_sum") + 64.0 * tot("TCC_EA0_RDREQ_64B_sum")
            wr = 1024.0 * tot("WRITE_SIZE")
            flat["_traffic"] = {"bytes_per_pass": (rd + wr) / passes, "read_bytes_per_pass": rd / passes, "write_bytes_per_pass": wr / passes,
                                "passes": passes, "fetch_size_x2_plus_write_size_bytes_per_pass": (2048.0 * tot("FETCH_SIZE") + wr) / passes,
                                "method": "128 x TCC_EA0_RDREQ_128B + 64 x TCC_EA0_RDREQ_64B + WRITE_SIZE (KiB), all kernels of the pass; beside it "
                                          "the guide's FETCH_SIZE x 2 + WRITE_SIZE"}
        flat["_command"] = "bench.py --workload %s --steps %d --warmup 1 --no-cpu %s" % (args.workload, args.steps, args.extra)
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(flat, open(args.out, "w"), indent=1)
        show = {c: round(v["mean_per_launch"], 1) for c, v in flat.items() if isinstance(v, dict) and "mean_per_launch" in v}
        print(args.workload, dom[:70], json.dumps(show))


if __name__ == "__main__":
    main()
