#!/bin/bash
# round 5, session C: parity of the round's new paths (phased hamming_search filter, device-driven levenshtein_exp rounds, unit-cost
# pre-pass option, checkpoint-and-recompute batch tracebacks), then their bench rows.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_trace.py tests/test_gpu_edge.py -x -q -m gpu > $O/tests_trace.txt 2>&1; tail -4 $O/tests_trace.txt
timeout 1500 python -m pytest tests/test_gpu_lev_batch.py -x -q -m gpu > $O/tests_batch.txt 2>&1; tail -4 $O/tests_batch.txt
timeout 1500 python -m pytest tests/test_gpu_search.py -x -q -m gpu -k "hamming" > $O/tests_ham.txt 2>&1; tail -4 $O/tests_ham.txt
run() { tag=$1; shift; timeout 600 python bench.py "$@" --no-cpu --no-pmc > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json")); print("$tag", round(d["ms_per_step"], 4), d["roofline"]["kernel_name"], round(d["value"], 1))
except Exception as e: print("$tag", "failed", e)
PY
}
run cfg2t --workload cfg2t --steps 5
TA_TUNING=1 TA_TRACE_TILE=32 run cfg2t_tile32 --workload cfg2t --steps 5
TA_TUNING=1 TA_TRACE_NO_BITS=1 run cfg2t_dp --workload cfg2t --steps 3
for n in 16 32 64; do run hsearch$n --workload hsearch --needle-len $n --steps 10; done
run cfg3_mutated --workload cfg3 --dist mutated --steps 3
TA_TUNING=1 TA_EXP_HOST_ROUNDS=1 run cfg3_mutated_host --workload cfg3 --dist mutated --steps 3
run cfg3 --workload cfg3 --steps 3
run cfg2w --workload cfg2w --steps 20
run cfg2w_prefilter --workload cfg2w --steps 20 --unit-prefilter
run cfg2w_mutated --workload cfg2w --dist mutated --steps 20
run cfg2w_mutated_prefilter --workload cfg2w --dist mutated --steps 20 --unit-prefilter
run cfg4w_prefilter --workload cfg4w --steps 20 --unit-prefilter
run cfg2 --steps 50
