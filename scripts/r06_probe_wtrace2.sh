#!/bin/bash
export TMPDIR=/tmp
cd /tmp
for sk in 0 1; do
  rm -rf /tmp/kt_w
  TA_TUNING=1 TA_TRACE_SKIP_EMIT=$sk rocprofv3 --kernel-trace --stats -d /tmp/kt_w -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2t --tcosts 2,3,1,- --tk 32 --steps 5 --warmup 1 --no-cpu --no-pmc --no-all-configs --prewarm-ms 0 > /dev/null 2>&1
  echo "skip_emit=$sk"; head -3 $(find /tmp/kt_w -name "kt_kernel_stats.csv" | head -1) | cut -c1-40,180-260
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_collect.py --out gpurun_out/r06/wtrace_pmc.json --workload cfg2t --sets sq1,sq2,rd_b,write --steps 5 --extra "--tcosts 2,3,1,- --tk 32 --prewarm-ms 0" 2>&1 | tail -2
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06/wtrace_pmc.json"))
for k,v in d["_kernels"].items():
    print(k[:50], {c:round(x["mean_per_launch"]) for c,x in v.items() if isinstance(x,dict)})
PY
