#!/bin/bash
# round 5, session G: the forward sweep of the checkpoint traceback folded into the distance pass (fixed-length batches), hamming_search's
# report through pinned memory: parity, then the rows.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_trace.py -x -q -m gpu > $O/tests_trace.txt 2>&1; tail -3 $O/tests_trace.txt
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_dist.py tests/test_gpu_rccl.py -x -q -m gpu > $O/tests_search.txt 2>&1; tail -3 $O/tests_search.txt
run() { tag=$1; shift; timeout 600 python bench.py "$@" --no-cpu --no-pmc > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json")); print("$tag", round(d["ms_per_step"], 4), d["roofline"]["kernel_name"], round(d["value"], 1))
except Exception as e: print("$tag", "failed", e)
PY
}
run cfg2t --workload cfg2t --steps 10
TA_TUNING=1 TA_TRACE_OWN_SWEEP=1 run cfg2t_own_sweep --workload cfg2t --steps 10
for n in 8 16 32 64; do run hsearch$n --workload hsearch --needle-len $n --steps 10; done
