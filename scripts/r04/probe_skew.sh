#!/bin/bash
# cfg4 with the wavefronts of a SIMD started out of phase (TA_BITS2_SKEW = s_sleep(16) units per phase step); 0 = off
O=gpurun_out/cfg4_skew; mkdir -p $O
export TA_TUNING=1
for sk in 0 1 4 16 64 0; do
  for n in 524288 1000000; do
    TA_BITS2_SKEW=$sk python bench.py --workload cfg4 --pairs $n --steps 50 --warmup 5 --no-cpu --no-pmc > $O/cfg4_${n}_skew$sk.json 2>/dev/null
    python -c "
import json; r=json.load(open('$O/cfg4_${n}_skew$sk.json')); print('skew', $sk, $n, r['ms_per_step'], r['roofline'].get('device_ms_per_pass'))"
  done
done
