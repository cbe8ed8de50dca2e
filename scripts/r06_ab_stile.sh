#!/bin/bash
# A/B: columns per fetch of the strings in the checkpoint trace kernel (TA_TRACE_STILE), cfg2tp
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="--workload cfg2tp --steps 20 --warmup 3 --no-cpu --no-pmc --no-all-configs"
for st in 32 64 128 64 128; do
  echo "STILE=$st $(TA_TUNING=1 TA_TRACE_STILE=$st python bench.py $B 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d["roofline"]["kernel_name"])')"
done
for st in 64 128; do
  TA_TUNING=1 TA_TRACE_STILE=$st python scripts/pmc_collect.py --out gpurun_out/r06/ab_stile${st}_pmc.json --workload cfg2tp --sets rd_b,write,issue --steps 5 --extra "--prewarm-ms 0" 2>&1 | tail -1
done
