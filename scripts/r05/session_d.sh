#!/bin/bash
# round 5, session D: after the walk's match-run fast path, the word-at-a-time run emission and the 8-step verdict windows: parity again,
# then cfg2t / hamming_search rows with their counter passes, and the unit pre-pass under the graph.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_trace.py tests/test_gpu_edge.py -x -q -m gpu > $O/tests_trace.txt 2>&1; tail -3 $O/tests_trace.txt
timeout 1500 python -m pytest tests/test_gpu_search.py -x -q -m gpu -k "hamming" > $O/tests_ham.txt 2>&1; tail -3 $O/tests_ham.txt
timeout 900 python -m pytest tests/test_gpu_lev_batch.py -x -q -m gpu -k "prefilter or device_driven" > $O/tests_batch.txt 2>&1; tail -3 $O/tests_batch.txt
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
for n in 12 16 24 32 48 64; do run hsearch$n --workload hsearch --needle-len $n --steps 10; done
run cfg2w_mutated_prefilter --workload cfg2w --dist mutated --steps 20 --unit-prefilter
run cfg2w_prefilter --workload cfg2w --steps 20 --unit-prefilter
(cd /tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2t --steps 3 --warmup 1 --no-cpu --no-pmc > /dev/null 2>&1; cp $(find /tmp/kt -name "kt_kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/cfg2t_kernel_stats.csv; head -6 $GRAFT_REPO_ROOT/$O/cfg2t_kernel_stats.csv | cut -c1-160)
python scripts/pmc_collect.py --out $O/hsearch32_pmc.json --workload hsearch --sets sq1,sq2 --steps 5 --extra "--needle-len 32" 2>&1 | tail -2
python scripts/pmc_collect.py --out $O/cfg2t_pmc.json --workload cfg2t --sets sq1,sq2,fetch,write,rd_b --steps 3 2>&1 | tail -2
