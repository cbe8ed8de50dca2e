#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lev_bits.py -x -q -k "widebits" 2>&1 | tail -2
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/kt3 -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/kt3.log 2>&1
head -5 $GRAFT_REPO_ROOT/gpurun_out/kt3/kt_kernel_stats.csv | cut -c1-120; tail -1 $GRAFT_REPO_ROOT/gpurun_out/kt3.log | cut -c1-200
