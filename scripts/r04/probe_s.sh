#!/bin/bash
# round-4 probe S: cfg5 pass with and without one agent-scope release / acquire fence per WORKGROUP of the wavefront exact kernel
mkdir -p gpurun_out/r04; O=gpurun_out/r04/probe_search_fence.txt; : > $O
for rep in 1 2; do
  echo "== default (release fence per workgroup before the done counter, acquire in the last workgroup)" >> $O
  python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu --no-pmc 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ms/step %.4f dev %.4f' % (r['ms_per_step'], r['roofline']['device_ms_per_pass']))" >> $O
  echo "== TA_SRCH_NO_FENCE=1 (relaxed agent-scope atomics + s_waitcnt only)" >> $O
  TA_TUNING=1 TA_SRCH_NO_FENCE=1 python bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu --no-pmc 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ms/step %.4f dev %.4f' % (r['ms_per_step'], r['roofline']['device_ms_per_pass']))" >> $O
done
echo "== search parts, default" >> $O; python scripts/measure_search_parts.py 2>&1 | tail -7 >> $O
echo "== search parts, TA_SRCH_NO_FENCE=1" >> $O; TA_TUNING=1 TA_SRCH_NO_FENCE=1 python scripts/measure_search_parts.py 2>&1 | tail -7 >> $O
cat $O
