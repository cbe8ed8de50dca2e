#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_lev_bits.py tests/test_gpu_lev_batch.py -x -q > gpurun_out/pytest_bits.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bits.log
tail -8 gpurun_out/pytest_bits.log
for ch in 32 64 16; do
echo "== chunk $ch"
TA_FORCE_CH=$ch timeout 600 python scripts/tune_band.py cfg2 0,0 2>&1 | grep GCUPS
TA_FORCE_CH=$ch timeout 600 python scripts/tune_band.py cfg4 0,0 2>&1 | grep GCUPS
done
TA_DEBUG=1 timeout 900 python bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu > gpurun_out/bench_cfg3.log 2>&1; tail -9 gpurun_out/bench_cfg3.log | cut -c1-400
