#!/bin/bash
# the small-alphabet kernels: tests, then cfg2 / cfg4 over dna / iupac / protein against the byte-test kernel on the same strings
mkdir -p gpurun_out/probe_q
python -m pytest tests/test_gpu_lev_bits.py -x -q -m gpu -k "alphabet" 2>&1 | tail -5
for d in dna iupac protein; do
  for w in cfg2 cfg4; do
    python bench.py --workload $w --dist $d --steps 20 --warmup 3 --no-cpu --no-pmc > gpurun_out/probe_q/${w}_$d.json 2> gpurun_out/probe_q/${w}_$d.err
    TA_NO_BITSQ=1 python bench.py --workload $w --dist $d --steps 20 --warmup 3 --no-cpu --no-pmc > gpurun_out/probe_q/${w}_${d}_bytetest.json 2>> gpurun_out/probe_q/${w}_$d.err
    python - <<PY
import json
for t in ("", "_bytetest"):
    try:
        r = json.load(open("gpurun_out/probe_q/${w}_$d%s.json" % t)); print("$w $d%s" % t, r["ms_per_step"], r["roofline"].get("kernel_name"), r["roofline"].get("device_ms_per_pass"))
    except Exception as e: print("$w $d%s" % t, "failed", e)
PY
  done
done
