#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "not search" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
python scripts/tune_band.py cfg2 0,0 22,3 24,3 34,2 66,1 18,4 12,6 8,9 16,5 28,3 40,2 > gpurun_out/tune_cfg2.log 2>&1
cat gpurun_out/tune_cfg2.log
python scripts/tune_band.py cfg4 0,0 18,1 20,1 10,2 6,3 4,5 12,2 22,1 24,1 > gpurun_out/tune_cfg4.log 2>&1
cat gpurun_out/tune_cfg4.log
