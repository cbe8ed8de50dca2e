#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_lev_bits.py tests/test_gpu_lev_batch.py tests/test_gpu_trace.py -x -q > gpurun_out/pytest_bits.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bits.log
tail -4 gpurun_out/pytest_bits.log
for ch in 32 64; do
echo "== chunk $ch"
TA_FORCE_CH=$ch timeout 600 python scripts/tune_band.py cfg2 0,0 2>&1 | grep GCUPS
TA_FORCE_CH=$ch timeout 600 python scripts/tune_band.py cfg4 0,0 2>&1 | grep GCUPS
TA_NO_BITS=1 TA_FORCE_CH=$ch timeout 600 python scripts/tune_band.py cfg2 0,0 2>&1 | grep GCUPS
TA_NO_BITS=1 TA_FORCE_CH=$ch timeout 600 python scripts/tune_band.py cfg4 0,0 2>&1 | grep GCUPS
done
