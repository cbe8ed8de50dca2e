#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SMEM"; do
rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4 -o p -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload ${WL:-cfg4} --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python - <<PY
import csv, collections, os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_c4/p_counter_collection.csv")))
agg=collections.defaultdict(list)
for r in rows:
    if 'lev_' in r['Kernel_Name'] and int(r['Grid_Size'])>100000: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v)/len(v)) for k,v in agg.items()}, rows[0]['Kernel_Name'][:50] if rows else '')
PY
done
