#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for w in 4 8 12 14; do
cd /tmp; TA_WB_WAVES_PER_CU=$w rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/kt3 -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/kt3.log 2>&1
echo "waves/CU $w: $(grep widebits $GRAFT_REPO_ROOT/gpurun_out/kt3/kt_kernel_stats.csv | cut -d, -f4)"
done
