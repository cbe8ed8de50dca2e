#!/bin/bash
# Round-end measurement pass: bench lines for every BASELINE config + rocprofv3 summaries for the default bench.
mkdir -p gpurun_out/final; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 900 python bench.py --workload cfg2 --dist mutated --no-cpu > $O/bench_cfg2_mutated.json 2>/dev/null
timeout 900 python bench.py --workload cfg4 --steps 20 --warmup 3 > $O/bench_cfg4.json 2>/dev/null
timeout 900 python bench.py --workload cfg1 --steps 50 --warmup 5 > $O/bench_cfg1.json 2>/dev/null
timeout 1500 python bench.py --workload cfg5 --steps 5 --warmup 2 > $O/bench_cfg5.json 2>/dev/null
timeout 1500 python bench.py --workload cfg3 --steps 3 --warmup 1 > $O/bench_cfg3.json 2>/dev/null
cd /tmp; rocprofv3 --kernel-trace --stats -d $O/kt3 -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu > $O/kt3.log 2>&1; cp $O/kt3/kt_kernel_stats.csv $O/bench_cfg3_kernel_stats.csv; rm -rf $O/kt3; cd $GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu > $O/kt.log 2>&1
cp $O/kt/kt_kernel_stats.csv $O/bench_cfg2_kernel_stats.csv
for set in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SMEM"; do
tag=$(echo $set | cut -d' ' -f1)
rocprofv3 --kernel-trace --pmc $set -d $O/pmc_cfg2_$tag -o p -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu > /dev/null 2>&1
done
# HBM-side traffic of every config's dominant kernels: FETCH_SIZE and WRITE_SIZE in separate passes
for wl in cfg1 cfg2 cfg3 cfg4 cfg5; do
for set in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $set -d $O/pmc_${wl}_$set -o p -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
done
done
python - <<PY
import csv, collections, os, json, glob, re
root=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/final"
out={}
for d in glob.glob(root+"/pmc_cfg2_SQ*")+glob.glob(root+"/pmc_cfg2_FETCH_SIZE")+glob.glob(root+"/pmc_cfg2_WRITE_SIZE"):
    rows=list(csv.DictReader(open(d+"/p_counter_collection.csv")))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'lev_bits' in r['Kernel_Name'] and int(r['Grid_Size'])>100000: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): out[k]={"mean_per_launch":sum(v)/len(v),"launches":len(v)}
json.dump(out,open(root+"/bench_cfg2_pmc.json","w"),indent=1)
# per config: sum over the path's kernels of one bench step (all launches / steps incl. warm-up and the parity call)
traffic={}
for wl in ("cfg1","cfg2","cfg3","cfg4","cfg5"):
    per={}
    for cnt in ("FETCH_SIZE","WRITE_SIZE"):
        rows=list(csv.DictReader(open(root+"/pmc_%s_%s/p_counter_collection.csv"%(wl,cnt))))
        byk=collections.defaultdict(list)
        for r in rows:
            if r['Kernel_Name'].startswith(('void ta::','ta::')): byk[re.sub(r'\(.*','',r['Kernel_Name'])].append(float(r['Counter_Value']))
        per[cnt]={k:(sum(v)/len(v),len(v)) for k,v in byk.items()}
    traffic[wl]=per
json.dump(traffic,open(root+"/traffic_raw.json","w"),indent=1)
print(json.dumps(traffic)[:1500])
PY
cd $GRAFT_REPO_ROOT; for f in gpurun_out/final/bench_cfg*.json; do echo $f; cut -c1-260 $f; done
head -3 gpurun_out/final/bench_cfg2_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/final/pmc_*/ gpurun_out/final/kt
