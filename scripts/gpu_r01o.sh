#!/bin/bash
# exp with the bag lower bound: parity tests + cfg3 bench with and without it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/o
timeout 600 python -m pytest tests/test_gpu_lev_batch.py tests/test_gpu_lev_wide.py tests/test_gpu_edge.py -x -q 2>&1 | tail -3
timeout 900 python bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu > gpurun_out/o/bench_cfg3.json 2>/dev/null
TA_EXP_NO_BOUND=1 timeout 900 python bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu > gpurun_out/o/bench_cfg3_nobound.json 2>/dev/null
TA_DEBUG=1 timeout 900 python bench.py --workload cfg3 --steps 1 --warmup 0 --no-cpu 2>&1 >/dev/null | grep "lev pass" | tail -4
for f in gpurun_out/o/*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2))
PY
done
