#!/bin/bash
# bit-parallel band kernel on cfg2: resident waves per CU (via block LDS) against time and L2-side traffic
mkdir -p gpurun_out/p; export TMPDIR=/tmp; cd /tmp
for lds in 0 45000 53000 65000 80000; do
  export TA_BITS_BLOCK_LDS=$lds
  t=$(python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['device_ms_per_pass'],4))")
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/p/f$lds -o p -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
  f=$(python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/p/f$lds/p_counter_collection.csv")) if 'lev_bits' in r['Kernel_Name']]
v=[float(r['Counter_Value']) for r in rows]
print(round(sum(v)/len(v)*2*1024/1e6), "MB", rows[0]['LDS_Block_Size'] if 'LDS_Block_Size' in rows[0] else '')
PY
)
  echo "block_lds=$lds ms=$t fetch=$f"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/p/f$lds
done
