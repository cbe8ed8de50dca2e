#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for wl in cfg2 cfg4 cfg1; do
  timeout 900 python bench.py --workload $wl --steps 10 --warmup 2 > gpurun_out/bench_$wl.log 2>&1; echo "rc=$?" >> gpurun_out/bench_$wl.log
  tail -2 gpurun_out/bench_$wl.log | cut -c1-1500
done
timeout 900 python bench.py --workload cfg2 --dist mutated --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_cfg2_mut.log 2>&1; tail -1 gpurun_out/bench_cfg2_mut.log | cut -c1-400
timeout 1500 python bench.py --workload cfg5 --steps 3 --warmup 1 > gpurun_out/bench_cfg5.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg5.log
tail -2 gpurun_out/bench_cfg5.log | cut -c1-1500
TA_DEBUG=1 timeout 1500 python bench.py --workload cfg3 --steps 2 --warmup 1 > gpurun_out/bench_cfg3.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg3.log
tail -14 gpurun_out/bench_cfg3.log | cut -c1-1500
