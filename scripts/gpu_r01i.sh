#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_kats.py -x -q > gpurun_out/pytest_search.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_search.log
tail -8 gpurun_out/pytest_search.log
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu 2>&1 | tail -1 | cut -c1-300
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/kt5 -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/kt5.log 2>&1
head -6 $GRAFT_REPO_ROOT/gpurun_out/kt5/kt_kernel_stats.csv | cut -c1-150
