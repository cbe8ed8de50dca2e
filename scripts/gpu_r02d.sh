#!/bin/bash
# round 2, GPU session d: line-form fetch vs chunk form (time, occupancy, L2 fabric-side requests), full suite, bench, latency, cell width
export TMPDIR=/tmp TA_TUNING=1
O=$GRAFT_REPO_ROOT/gpurun_out/r02d; mkdir -p $O
cd $GRAFT_REPO_ROOT
python scripts/exp_memory_bound.py TA_BITS_NO_COOP=1 TA_BITS_BLOCK_LDS=53000 TA_BITS_BLOCK_LDS=53000,TA_BITS_NO_COOP=1 \
  TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=10200 TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=10900 TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=12300 TA_BITS_WPB=2 "" TA_BITS_NO_COOP=1 "" 2>&1 | grep -v Warning > $O/memory_bound.txt; cat $O/memory_bound.txt
python scripts/pmc_collect.py --out $O/traffic_cfg2_line.json --workload cfg2 --sets fetch,write,rd_b,hit,req --steps 5 2>&1 | tail -1
TA_BITS_NO_COOP=1 python scripts/pmc_collect.py --out $O/traffic_cfg2_chunk.json --workload cfg2 --sets fetch,rd_b --steps 5 2>&1 | tail -1
python scripts/pmc_collect.py --out $O/traffic_cfg4_line.json --workload cfg4 --sets fetch,rd_b --steps 5 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
python scripts/measure_latency.py > $O/latency.txt 2>&1; cat $O/latency.txt
./scripts/ubench_cellwidth > $O/ubench_cellwidth.txt 2>&1; cat $O/ubench_cellwidth.txt
timeout 600 python bench.py > $O/bench_cfg2.json 2>$O/bench_cfg2.err; cut -c1-400 $O/bench_cfg2.json
timeout 600 python bench.py --workload cfg4 --no-cpu --steps 50 > $O/bench_cfg4.json 2>/dev/null; cut -c1-300 $O/bench_cfg4.json
