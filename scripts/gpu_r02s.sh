#!/bin/bash
# session s: what bounds the stride-8 kernel at 0.315 ms -- counters of s8 / s8b and the no-HBM bound
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
cp triple_accel_amd/libtriple_accel_amd.so /tmp/ta_keep.so
for so in s8 s8b; do
  cp ab/$so.so triple_accel_amd/libtriple_accel_amd.so
  python scripts/exp_memory_bound.py > $O/membound_$so.txt 2>&1; python scripts/exp_memory_bound.py >> $O/membound_$so.txt 2>&1
  grep -v amdgpu $O/membound_$so.txt
  python scripts/pmc_collect.py --out $O/pmc_cfg2_$so.json --workload cfg2 --sets sq1,sq2 --steps 5 2>&1 | tail -1 | cut -c1-900
done
cp /tmp/ta_keep.so triple_accel_amd/libtriple_accel_amd.so
