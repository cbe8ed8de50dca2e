#!/bin/bash
# A/B of the small-alphabet kernels against the byte-test kernels on the same strings: raw bench lines under gpurun_out/ab_alphabet/
O=gpurun_out/ab_alphabet; mkdir -p $O
export TA_TUNING=1
for d in dna dna5 iupac protein; do
  for w in cfg2 cfg4; do
    TA_BITSQ_WIDE=$([ $d = dna ] && echo 0 || echo 1) python bench.py --workload $w --dist $d --steps 50 --warmup 5 --no-cpu --no-pmc > $O/${w}_${d}_table.json 2>/dev/null
    TA_NO_BITSQ=1 python bench.py --workload $w --dist $d --steps 50 --warmup 5 --no-cpu --no-pmc > $O/${w}_${d}_bytetest.json 2>/dev/null
    python bench.py --workload $w --dist $d --steps 50 --warmup 5 --no-cpu --no-pmc > $O/${w}_${d}_default.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/ab_alphabet/*.json")):
    try:
        r = json.load(open(f)); print(os.path.basename(f), r["ms_per_step"], r["roofline"]["kernel_name"])
    except Exception as e: print(f, "failed", e)
PY
