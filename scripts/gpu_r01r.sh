#!/bin/bash
# after the v_perm byte test in the bit-parallel band kernel: parity + cfg2 / cfg4 / cfg3 timings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r
timeout 900 python -m pytest tests/test_gpu_lev_bits.py tests/test_gpu_lev_batch.py tests/test_gpu_kats.py -x -q 2>&1 | tail -2
for wl in cfg2 cfg4; do timeout 600 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu 2>/dev/null > gpurun_out/r/bench_$wl.json; done
timeout 600 python bench.py --workload cfg2 --dist mutated --steps 50 --warmup 5 --no-cpu 2>/dev/null > gpurun_out/r/bench_cfg2_mutated.json
TA_EXP_NO_BOUND=1 timeout 600 python bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu 2>/dev/null > gpurun_out/r/bench_cfg3_nobound.json
for f in gpurun_out/r/*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],4), round(d["roofline"]["frac"],4))
PY
done
