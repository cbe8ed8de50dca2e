#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_lev_bits.py tests/test_gpu_lev_batch.py -x -q 2>&1 | tail -3
timeout 600 python scripts/tune_band.py cfg2 0,0 2>&1 | grep GCUPS
timeout 600 python scripts/tune_band.py cfg4 0,0 2>&1 | grep GCUPS
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_fetch -o f -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python - <<PY
import csv, os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_fetch/f_counter_collection.csv")))
v=[float(r['Counter_Value']) for r in rows if 'lev_bits' in r['Kernel_Name']]
print("lev_bits FETCH_SIZE KiB/launch (raw): mean %.0f  -> x2 = %.0f MB (algorithmic 512 MB)" % (sum(v)/len(v), 2*1024*sum(v)/len(v)/1e6))
PY
