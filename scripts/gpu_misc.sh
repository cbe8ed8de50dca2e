#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/measure_misc.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/measure_misc.log
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/prof_cal_$c -o cal -f csv -- python $GRAFT_REPO_ROOT/scripts/measure_misc.py > /dev/null 2>&1
python - <<PY
import csv, collections, os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_cal_$c/cal_counter_collection.csv")))
agg=collections.defaultdict(list)
for r in rows: agg[r['Kernel_Name'][:50]].append(float(r['Counter_Value']))
for k,v in agg.items(): print("$c", k, "mean KiB/launch = %.0f over %d launches" % (sum(v)/len(v), len(v)))
PY
done
