#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python scripts/measure_misc.py 2>&1 | grep -v amdgpu.ids | tail -2
python scripts/tune_band.py cfg2 0,0 22,3 2>&1 | tail -2
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_fetch -o f -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python - <<PY
import csv, collections, os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_fetch/f_counter_collection.csv")))
v=[float(r['Counter_Value']) for r in rows if 'lev_band' in r['Kernel_Name']]
print("lev_band FETCH_SIZE KiB/launch (raw): mean %.0f  -> x2 = %.0f MB (algorithmic 512 MB)" % (sum(v)/len(v), 2*1024*sum(v)/len(v)/1e6))
PY
cd $GRAFT_REPO_ROOT; bash scripts/gpu_tests.sh tests/test_gpu_lev_batch.py 2>&1 | tail -3
