#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python bench.py --workload cfg5 --steps 3 --warmup 1 > gpurun_out/bench_cfg5.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg5.log
tail -2 gpurun_out/bench_cfg5.log | cut -c1-1800
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg3 -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_cfg3.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof_cfg3/kt_kernel_stats.csv | cut -c1-200
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_cfg3/kt_kernel_trace.csv")))
rows=[r for r in rows if 'lev_' in r['Kernel_Name']]
for r in rows[-9:]:
    print(r['Kernel_Name'][:60], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, "ms", r.get('Grid_Size'))
PY
