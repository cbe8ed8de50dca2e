#!/bin/bash
# session k: where cfg4's 14 % of idle issue slots go -- occupancy / waves per block sweep and the no-HBM (stride 0) bound
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
export TA_TUNING=1
EXP_WL=cfg4 python scripts/exp_memory_bound.py "" TA_BITS_WPB=1 TA_BITS_WPB=2 TA_BITS_BLOCK_LDS=53000 TA_BITS_BLOCK_LDS=80000 \
   TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=11000 TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=12500 TA_BITS_WPB=2,TA_BITS_BLOCK_LDS=23000 "" > gpurun_out/r02k/cfg4_sweep.txt 2>&1
cat gpurun_out/r02k/cfg4_sweep.txt
