#!/bin/bash
# round 2, GPU session c: memory share of the band kernel, occupancy / block-shape sweep, coalesced vs per-lane fetch, tests
export TMPDIR=/tmp TA_TUNING=1
O=$GRAFT_REPO_ROOT/gpurun_out/r02c; mkdir -p $O
cd $GRAFT_REPO_ROOT
python scripts/exp_memory_bound.py TA_BITS_NO_COOP=1 TA_BITS_BLOCK_LDS=40000 TA_BITS_BLOCK_LDS=40000,TA_BITS_NO_COOP=1 \
  TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=12300 TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=11600 TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=10900 TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=10200 \
  TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=13600 TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=14800 TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=16300 TA_BITS_WPB=2,TA_BITS_BLOCK_LDS=26600 \
  TA_BITS_NO_COOP=1 "" 2>&1 | grep -v Warning > $O/memory_bound.txt; cat $O/memory_bound.txt
python scripts/pmc_collect.py --out $O/traffic_cfg2_coop.json --workload cfg2 --sets fetch,rd_b,hit,req --steps 5 2>&1 | tail -1
TA_BITS_NO_COOP=1 python scripts/pmc_collect.py --out $O/traffic_cfg2_perlane.json --workload cfg2 --sets fetch,rd_b,hit,req --steps 5 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
python scripts/measure_latency.py > $O/latency.txt 2>&1; cat $O/latency.txt
