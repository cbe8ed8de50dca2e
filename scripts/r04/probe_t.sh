#!/bin/bash
# A/B of the fused v_bfi tree (one asm statement): cfg2 / cfg4 / ragged / cfg2s timings + the bit-parallel kernels' GPU tests
mkdir -p gpurun_out/probe_t
python -m pytest tests/test_gpu_lev_bits.py tests/test_gpu_kats.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
for w in "cfg2" "cfg4" "cfg2 --dist ragged" "cfg2s"; do
  t=$(echo $w | tr -d ' -')
  python bench.py --workload $w --steps 100 --warmup 5 --no-cpu --no-pmc > gpurun_out/probe_t/${t}_$i.json 2>/dev/null
  python -c "
import json; r=json.load(open('gpurun_out/probe_t/${t}_$i.json')); print('$w', r['ms_per_step'], r['roofline'].get('device_ms_per_pass'), r['roofline']['kernel_name'][:40])"
done; done
