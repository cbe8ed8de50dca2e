#!/bin/bash
# Round 5: the rows that moved after the main pass (scripts/r05/gpu_profiles_r05.sh) -- cfg2t with the forward sweep folded into the distance
# pass (cfg2t_own_sweep: TA_TRACE_OWN_SWEEP=1, an A/B row) and the hamming_search rows with the report through pinned memory -- same files, same box.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $O
cd $GRAFT_REPO_ROOT
flags() { case $1 in hsearch8) echo "--workload hsearch --needle-len 8" ;; hsearch16) echo "--workload hsearch --needle-len 16" ;; hsearch32) echo "--workload hsearch --needle-len 32" ;;
  hsearch64) echo "--workload hsearch --needle-len 64" ;; cfg2t_own_sweep) echo "--workload cfg2t" ;; *) echo "--workload $1" ;; esac; }
envof() { case $1 in cfg2t_own_sweep) echo "TA_TUNING=1 TA_TRACE_OWN_SWEEP=1" ;; *) echo "TA_NOENV=1" ;; esac; }
for tag in cfg2t cfg2t_own_sweep hsearch8 hsearch16 hsearch32 hsearch64; do
  env $(envof $tag) timeout 900 python bench.py $(flags $tag) --steps 10 --warmup 2 --no-cpu --no-pmc > $O/bench_$tag.json 2> $O/bench_$tag.err
  [ $tag = cfg2t_own_sweep ] && continue
  (cd /tmp; rm -rf /tmp/kt_$tag; rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py $(flags $tag) --steps 5 --warmup 1 --no-cpu --no-pmc 2>/dev/null | grep '^{' > $O/bench_${tag}_under_kernel_trace.json; cp $(find /tmp/kt_$tag -name "kt_kernel_stats.csv" | head -1) $O/bench_${tag}_kernel_stats.csv; rm -rf /tmp/kt_$tag)
  wl=$(flags $tag | cut -d' ' -f2); extra=$(flags $tag | cut -s -d' ' -f3-)
  python scripts/pmc_collect.py --out $O/bench_${tag}_pmc.json --workload $wl --sets sq1,sq2,fetch,write,rd_b --steps 5 --extra "$extra" 2>&1 | tail -1 | cut -c1-120
done
for f in $O/bench_cfg2t.json $O/bench_cfg2t_own_sweep.json $O/bench_hsearch8.json $O/bench_hsearch16.json $O/bench_hsearch32.json $O/bench_hsearch64.json; do python - <<PY
import json
d = json.load(open("$f")); print("$f".split("/")[-1], round(d["ms_per_step"], 4), d["roofline"]["kernel_name"])
PY
done
