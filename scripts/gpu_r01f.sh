#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_lev_bits.py tests/test_gpu_lev_wide.py -x -q > gpurun_out/pytest_bits.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bits.log
tail -12 gpurun_out/pytest_bits.log
TA_DEBUG=1 timeout 900 python bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu > gpurun_out/bench_cfg3.log 2>&1; tail -7 gpurun_out/bench_cfg3.log | cut -c1-330
