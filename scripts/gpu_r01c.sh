#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for ch in 64 32 16; do
echo "== chunk $ch"
TA_FORCE_CH=$ch timeout 600 python scripts/tune_band.py cfg2 34,1 18,2 2>&1 | grep GCUPS
TA_FORCE_CH=$ch timeout 600 python scripts/tune_band.py cfg4 10,1 6,2 2>&1 | grep GCUPS
done
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
