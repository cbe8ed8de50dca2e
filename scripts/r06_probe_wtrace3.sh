#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_trace.py tests/test_gpu_adversarial.py tests/test_gpu_lev_batch.py -q -x 2>&1 | tail -4
B="--steps 10 --warmup 2 --no-cpu --no-pmc --no-all-configs"
for args in "--tcosts 2,3,1,- --tk 32" "--tcosts 2,3,1,- --tk 64" "--tcosts 2,2,1,3 --tk 32" "--tcosts 2,3,0,- --tk 32"; do
  echo "$args: $(python bench.py --workload cfg2t $args $B 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d["roofline"]["kernel_name"], d["kernel"]["diags_per_lane"])')"
done
echo "unit costs through the DP route: $(TA_TUNING=1 TA_TRACE_NO_BITS=1 python bench.py --workload cfg2t $B 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d["roofline"]["kernel_name"])')"
python scripts/pmc_collect.py --out gpurun_out/r06/wtrace_pmc2.json --workload cfg2t --sets issue,rd_b,write --steps 5 --extra "--tcosts 2,3,1,- --tk 32 --prewarm-ms 0" 2>&1 | tail -1
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06/wtrace_pmc2.json"))
for k,v in d["_kernels"].items():
    print(k[:50], {c:round(x["mean_per_launch"]) for c,x in v.items() if isinstance(x,dict)})
PY
