#!/bin/bash
# Round 5's measurement pass on ONE GPU box (scripts/gpu_profiles.sh with this round's rows): every number quoted in README.md / DESIGN.md
# comes from the files this writes (copy gpurun_out/profiles/* into profiles/r05/, then scripts/make_tables.py r05).
# New rows: cfg5w_* (levenshtein_search under weighted EditCosts through the superset filter), cfg3_mutated / cfg4_mutated (similar strings:
# the levenshtein_exp rounds, device-driven), cfg2w_prefilter / cfg2w_mutated_prefilter / cfg4w_prefilter (TA_OPT_UNIT_PREFILTER), cfg2t on
# the checkpoint-and-recompute kernel and on the DP kernel's records (cfg2t_dp), hsearch16 / 64 on the phased bit-sliced filter.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $O
cd $GRAFT_REPO_ROOT
(lscpu | head -25; echo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; rocminfo | grep -E "Marketing|Compute Unit|Max Clock|gfx" | head -12) > $O/host_cpu.txt 2>&1
flags() {   # tag -> bench.py flags
  case $1 in
    cfg2_mutated) echo "--workload cfg2 --dist mutated" ;;
    cfg3_mutated) echo "--workload cfg3 --dist mutated" ;;
    cfg3_mutated_host_rounds) echo "--workload cfg3 --dist mutated" ;;
    cfg4_mutated) echo "--workload cfg4 --dist mutated" ;;
    cfg2_ragged) echo "--workload cfg2 --dist ragged" ;;
    cfg2_dna) echo "--workload cfg2 --dist dna" ;;
    cfg2_dna5) echo "--workload cfg2 --dist dna5" ;;
    cfg2t_dp) echo "--workload cfg2t" ;;
    cfg2w_prefilter) echo "--workload cfg2w --unit-prefilter" ;;
    cfg2w_mutated) echo "--workload cfg2w --dist mutated" ;;
    cfg2w_mutated_prefilter) echo "--workload cfg2w --dist mutated --unit-prefilter" ;;
    cfg4w_prefilter) echo "--workload cfg4w --unit-prefilter" ;;
    cfg5w_220) echo "--workload cfg5w --costs 2,2,0,-" ;;
    cfg5w_231) echo "--workload cfg5w --costs 2,3,1,-" ;;
    cfg5w_2213) echo "--workload cfg5w --costs 2,2,1,3" ;;
    cfg5w_1101) echo "--workload cfg5w --costs 1,1,0,1" ;;
    cfg5w_231_nofilter) echo "--workload cfg5w --costs 2,3,1,-" ;;
    hsearch8) echo "--workload hsearch --needle-len 8" ;;
    hsearch16) echo "--workload hsearch --needle-len 16" ;;
    hsearch32) echo "--workload hsearch --needle-len 32" ;;
    hsearch64) echo "--workload hsearch --needle-len 64" ;;
    hsearch32_r04) echo "--workload hsearch --needle-len 32" ;;
    hsearch64_r04) echo "--workload hsearch --needle-len 64" ;;
    *) echo "--workload $1" ;;
  esac
}
steps() { case $1 in cfg3) echo "--steps 3 --warmup 1" ;; cfg3_mutated*) echo "--steps 10 --warmup 2" ;; cfg5*|hsearch*|cfg2t*) echo "--steps 10 --warmup 2" ;; cfg2) echo "" ;; *) echo "--steps 50" ;; esac; }
envof() { case $1 in cfg2t_dp) echo "TA_TUNING=1 TA_TRACE_NO_BITS=1" ;; cfg3_mutated_host_rounds) echo "TA_TUNING=1 TA_EXP_HOST_ROUNDS=1" ;; cfg5w_231_nofilter) echo "TA_TUNING=1 TA_SEARCH_NOWFILTER=1" ;;
                    hsearch*_r04) echo "TA_TUNING=1 TA_HAMMING_SEARCH_NO_PHASE=1" ;; *) echo "TA_NOENV=1" ;; esac; }
TAGS="cfg2 cfg2_mutated cfg4 cfg4_mutated cfg1 cfg5 cfg3 cfg3_mutated cfg2w cfg4w cfg2l cfg2s cfg2t cfg2_ragged cfg2_dna cfg2_dna5 hsearch8 hsearch16 hsearch32 hsearch64 cfg5w_220 cfg5w_231 cfg5w_2213 cfg5w_1101 cfg2w_prefilter cfg2w_mutated cfg2w_mutated_prefilter cfg4w_prefilter"
AB="cfg2t_dp cfg3_mutated_host_rounds cfg5w_231_nofilter hsearch32_r04 hsearch64_r04"       # A/B rows: bench line only
for tag in $TAGS $AB; do
  nocpu="--no-cpu --no-pmc"; [ $tag = cfg2 ] && nocpu=""       # (cfg2: the driver's command -- cpu_baseline leg and the live counter passes included)
  env $(envof $tag) timeout 900 python bench.py $(flags $tag) $(steps $tag) $nocpu > $O/bench_$tag.json 2> $O/bench_$tag.err
done
timeout 600 python bench.py --early-out --no-cpu --no-pmc > $O/bench_cfg2_early_out.json 2>/dev/null
timeout 600 python bench.py --pairs 2000000 --no-cpu --no-pmc > $O/bench_cfg2_2m.json 2>/dev/null
for tag in $TAGS; do
  case $tag in cfg2_mutated|cfg2w_mutated|cfg4w_prefilter|cfg5w_220|cfg5w_1101) continue ;; esac     # (bench line only)
  st=5; [ $tag = cfg3 ] && st=3
  (cd /tmp; rm -rf /tmp/kt_$tag; env $(envof $tag) rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py $(flags $tag) --steps $st --warmup 1 --no-cpu --no-pmc 2>/dev/null | grep '^{' > $O/bench_${tag}_under_kernel_trace.json; cp $(find /tmp/kt_$tag -name "kt_kernel_stats.csv" | head -1) $O/bench_${tag}_kernel_stats.csv; rm -rf /tmp/kt_$tag)
  wl=$(flags $tag | cut -d' ' -f2); extra=$(flags $tag | cut -s -d' ' -f3-)
  env $(envof $tag) python scripts/pmc_collect.py --out $O/bench_${tag}_pmc.json --workload $wl --sets sq1,sq2,fetch,write,rd_b --steps $st --extra "$extra" 2>&1 | tail -1
done
python scripts/measure_latency.py > $O/latency.txt 2>&1
./scripts/ubench_mix > $O/ubench_mix.txt 2>&1
for f in $O/bench_*.json; do case $f in *_pmc.json|*_under_kernel_trace.json) continue ;; esac; python - <<PY
import json
try:
    d = json.load(open("$f")); print("$f".split("/")[-1], round(d["ms_per_step"], 4), d["roofline"]["kernel_name"])
except Exception as e: print("$f", "failed", e)
PY
done
