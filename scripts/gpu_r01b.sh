#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
timeout 600 python scripts/tune_band.py cfg2 0,0 34,1 18,2 12,3 40,1 > gpurun_out/tune_cfg2.log 2>&1; cat gpurun_out/tune_cfg2.log | tail -8
timeout 600 python scripts/tune_band.py cfg4 0,0 10,1 12,1 6,2 4,3 2,5 > gpurun_out/tune_cfg4.log 2>&1; cat gpurun_out/tune_cfg4.log | tail -8
TA_DEBUG=1 timeout 900 python bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu > gpurun_out/bench_cfg3.log 2>&1; tail -12 gpurun_out/bench_cfg3.log | cut -c1-600
