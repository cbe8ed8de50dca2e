#!/bin/bash
# round 5, session E: string tiles of 64 columns in the checkpoint trace kernel; the capture-aware stream guard (bench --unit-prefilter under its graph)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_trace.py -x -q -m gpu > $O/tests_trace.txt 2>&1; tail -3 $O/tests_trace.txt
run() { tag=$1; shift; timeout 600 python bench.py "$@" --no-cpu --no-pmc > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json")); print("$tag", round(d["ms_per_step"], 4), d["roofline"]["kernel_name"], round(d["value"], 1))
except Exception as e: print("$tag", "failed", e)
PY
}
run cfg2t --workload cfg2t --steps 5
run cfg2w_mutated_prefilter --workload cfg2w --dist mutated --steps 20 --unit-prefilter
run cfg2w_prefilter --workload cfg2w --steps 20 --unit-prefilter
python scripts/pmc_collect.py --out $O/cfg2t_pmc.json --workload cfg2t --sets sq1,sq2,rd_b --steps 3 2>&1 | tail -1
