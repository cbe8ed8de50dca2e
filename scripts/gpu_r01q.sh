#!/bin/bash
# 12 vs 16 resident waves per CU for the bit-parallel band kernel: cfg2 / cfg4 time and L2-side traffic, twice each
mkdir -p gpurun_out/q; export TMPDIR=/tmp; cd /tmp
for wl in cfg2 cfg4; do for lds in 53000 38912 53000 38912; do
  export TA_BITS_BLOCK_LDS=$lds
  t=$(python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 50 --warmup 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4))")
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/q/f -o p -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
  f=$(python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/q/f/p_counter_collection.csv")) if 'lev_bits' in r['Kernel_Name']]
v=[float(r['Counter_Value']) for r in rows]
print(round(sum(v)/len(v)*2*1024/1e6), "MB")
PY
)
  echo "$wl block_lds=$lds ms=$t fetch=$f"; rm -rf $GRAFT_REPO_ROOT/gpurun_out/q/f
done; done
