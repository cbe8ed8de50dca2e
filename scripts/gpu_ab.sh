#!/bin/bash
# A/B on ONE box: bench the library built from HEAD~N (gpurun_out-independent copy under /tmp) against the working tree's.
# usage: gpu_ab.sh <old .so path relative to the repo> ; prints device ms per pass for cfg2 / cfg4, alternating three times
cd $GRAFT_REPO_ROOT
OLD=$1
cp triple_accel_amd/libtriple_accel_amd.so /tmp/new.so
for rep in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp $OLD triple_accel_amd/libtriple_accel_amd.so; else cp /tmp/new.so triple_accel_amd/libtriple_accel_amd.so; fi
    for wl in cfg2 cfg4; do
      t=$(python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4))")
      echo "$v $wl $t"
    done
  done
done
cp /tmp/new.so triple_accel_amd/libtriple_accel_amd.so
