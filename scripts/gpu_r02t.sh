#!/bin/bash
# session t: stride-8 kernel -- occupancy / block shape sweep, counter list
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp TA_TUNING=1
O=gpurun_out/r02t; mkdir -p $O
python scripts/exp_memory_bound.py "" TA_BITS_WPB=1 TA_BITS_WPB=2 TA_BITS_BLOCK_LDS=53000 TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=11000 TA_BITS_WPB=1,TA_BITS_BLOCK_LDS=10500 TA_BITS_WPB=2,TA_BITS_BLOCK_LDS=23000 "" > $O/cfg2_s8_sweep.txt 2>&1
grep -v amdgpu $O/cfg2_s8_sweep.txt
(cd /tmp; rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TCC|TD|GRBM|SPI)_[A-Z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/$O/counters.txt)
wc -l $O/counters.txt; grep -E "WAIT|STALL|BUSY|VMEM|LEVEL" $O/counters.txt | tr '\n' ' '
