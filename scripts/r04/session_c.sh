#!/bin/bash
# round-4 GPU session C: default ragged path (chunk form, 8-byte classes, longest first), unit costs times g, hipGraph timed region
mkdir -p gpurun_out/r04; O=gpurun_out/r04
python -m pytest tests/test_gpu_lev_batch.py tests/test_gpu_bench.py tests/test_gpu_rccl.py -x -q 2>&1 | tail -6 > $O/t_c.txt
for w in "cfg2" "cfg2 --no-graph" "cfg2 --dist ragged" "cfg2l" "cfg2s" "cfg2w" "cfg4" "cfg4 --no-graph"; do
  tag=$(echo $w | tr -d ' -'); python bench.py --workload $w --steps 50 --no-cpu > $O/c_$tag.json 2> $O/c_$tag.err
  python - <<PY
import json
try:
    r=json.load(open("$O/c_$tag.json")); print("%-22s value %9.1f  ms/step %.4f  dev %.4f  %s  %s" % ("$w", r["value"], r["ms_per_step"], r["roofline"]["device_ms_per_pass"], r.get("timed_region"), r["roofline"]["kernel_name"]))
except Exception as e: print("$w FAILED", e, open("$O/c_$tag.err").read()[-600:])
PY
done
python bench.py --steps 20 --warmup 5 > $O/c_driver_cmd.json 2> $O/c_driver_cmd.err; python -c "
import json; r=json.load(open('$O/c_driver_cmd.json')); print('driver protocol (20/5, cpu leg): value %.1f ms/step %.4f dev %.4f %s' % (r['value'], r['ms_per_step'], r['roofline']['device_ms_per_pass'], r.get('timed_region')))"
cat $O/t_c.txt
