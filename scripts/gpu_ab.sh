#!/bin/bash
# A/B of library builds on ONE box (boxes differ by +- 3 %): device ms per pass for cfg2 / cfg4, alternating REPS times.
# usage: [AB_WORKLOADS="cfg2 cfg4"] [AB_FLAGS="--dist dna"] gpu_ab.sh [REPS=3] <a.so> <b.so> ...   (paths relative to the repo; the
# working tree's library is restored at the end)
cd $GRAFT_REPO_ROOT
REPS=3
if [[ $1 =~ ^[0-9]+$ ]]; then REPS=$1; shift; fi
cp triple_accel_amd/libtriple_accel_amd.so /tmp/ta_keep.so
for rep in $(seq $REPS); do
  for so in "$@"; do
    cp $so triple_accel_amd/libtriple_accel_amd.so
    for wl in ${AB_WORKLOADS:-cfg2 cfg4}; do
      t=$(python bench.py --workload $wl $AB_FLAGS --steps 100 --warmup 10 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4), round(d['ms_per_step'],4))")
      echo "$(basename $so) $wl $t"
    done
  done
done
cp /tmp/ta_keep.so triple_accel_amd/libtriple_accel_amd.so
