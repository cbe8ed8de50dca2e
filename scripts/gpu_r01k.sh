#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for f in 64 32; do
cd /tmp; TA_FORCE_WIDEBITS=$f rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/kt3 -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/kt3.log 2>&1
echo "rows/lane $f"; head -3 $GRAFT_REPO_ROOT/gpurun_out/kt3/kt_kernel_stats.csv | cut -c1-120
done
