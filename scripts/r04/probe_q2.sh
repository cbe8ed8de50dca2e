#!/bin/bash
# quick timing of the wide-alphabet kernel: cfg2 / cfg4 over iupac / protein (+ the test of the kernel)
mkdir -p gpurun_out/probe_q
for d in dna5 iupac protein; do
  for w in cfg2 cfg4; do
    python bench.py --workload $w --dist $d --steps 20 --warmup 3 --no-cpu --no-pmc > gpurun_out/probe_q/${w}_$d.json 2> gpurun_out/probe_q/${w}_$d.err
    TA_TUNING=1 TA_NO_BITSQW=1 python bench.py --workload $w --dist $d --steps 20 --warmup 3 --no-cpu --no-pmc > gpurun_out/probe_q/${w}_${d}_bytetest.json 2>> gpurun_out/probe_q/${w}_$d.err
    python -c "
import json; r = json.load(open('gpurun_out/probe_q/${w}_${d}_bytetest.json')); print('$w $d bytetest', r['ms_per_step'], r['roofline'].get('kernel_name'))"
    python - <<PY
import json
try:
    r = json.load(open("gpurun_out/probe_q/${w}_$d.json")); print("$w $d", r["ms_per_step"], r["roofline"].get("kernel_name"), r["roofline"].get("device_ms_per_pass"))
except Exception as e: print("$w $d", "failed", e)
PY
  done
done
