#!/bin/bash
# Round 5, late: cfg2t again after the trace kernel's prefetches (the next tile's checkpoint and the next string tile one tile ahead) --
# bench line, kernel-trace stats and counter passes into the same files as scripts/r05/gpu_profiles_r05_refresh.sh.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $O
cd $GRAFT_REPO_ROOT
tag=cfg2t
TA_NOENV=1 timeout 900 python bench.py --workload cfg2t --steps 10 --warmup 2 --no-cpu --no-pmc > $O/bench_$tag.json 2> $O/bench_$tag.err
(cd /tmp; rm -rf /tmp/kt_$tag; rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2t --steps 5 --warmup 1 --no-cpu --no-pmc 2>/dev/null | grep '^{' > $O/bench_${tag}_under_kernel_trace.json; cp $(find /tmp/kt_$tag -name "kt_kernel_stats.csv" | head -1) $O/bench_${tag}_kernel_stats.csv; rm -rf /tmp/kt_$tag)
python scripts/pmc_collect.py --out $O/bench_${tag}_pmc.json --workload cfg2t --sets sq1,sq2,fetch,write,rd_b --steps 5 2>&1 | tail -1 | cut -c1-120
head -4 $O/bench_${tag}_kernel_stats.csv | cut -c1-200
python - <<PY
import json
d = json.load(open("$O/bench_cfg2t.json")); print("cfg2t", round(d["ms_per_step"], 4), d["roofline"]["kernel_name"])
PY
