#!/bin/bash
# counters + kernel time of the pair-sliced band kernel on the cfg2 bench
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_sl -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > /dev/null 2>&1
grep -E "lev_sliced|lev_bits" $GRAFT_REPO_ROOT/gpurun_out/prof_sl/kt_kernel_stats.csv | cut -c1-200
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/prof_sl -o p -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python - <<PY
import csv, collections, os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_sl/p_counter_collection.csv")))
agg=collections.defaultdict(list)
for r in rows:
    if 'lev_sliced' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
done
