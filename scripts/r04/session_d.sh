#!/bin/bash
mkdir -p gpurun_out/r04; O=gpurun_out/r04
python scripts/r04/dbg_scale.py 2>&1 | tail -12
for f in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --no-graph" "--steps 20 --warmup 50" "--steps 50 --warmup 5" "--steps 100 --warmup 5" "--steps 20 --warmup 5 --prewarm-ms 1000"; do
  python bench.py $f --no-cpu > $O/d.json 2>/dev/null; python -c "
import json; r=json.load(open('$O/d.json')); print('%-42s ms/step %.4f dev %.4f ramp %d %s' % ('$f', r['ms_per_step'], r['roofline']['device_ms_per_pass'], r['prewarm_passes'], r.get('timed_region')))"
done
