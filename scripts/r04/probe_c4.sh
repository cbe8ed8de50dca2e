#!/bin/bash
# cfg4 by batch size: what of the pass is per resident set of wavefronts and what is fixed (profiles/r04/raw/cfg4_sizes/)
O=gpurun_out/cfg4_sizes; mkdir -p $O
for n in 262144 524288 1000000 1048576 1572864 2097152 4194304; do
  python bench.py --workload cfg4 --pairs $n --steps 50 --warmup 5 --no-cpu --no-pmc > $O/cfg4_$n.json 2>/dev/null
  python -c "
import json; r=json.load(open('$O/cfg4_$n.json')); print($n, r['ms_per_step'], r['roofline'].get('device_ms_per_pass'), r['roofline']['kernel_name'][:40])"
done
