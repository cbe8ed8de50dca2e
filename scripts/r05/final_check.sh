#!/bin/bash
# Round 5, last build: every profiled workload's bench line once more (no counters, no CPU leg) next to the committed profile of the
# same workload -- a regression check after the late changes, not a new set of numbers (boxes differ by +- 3-5 %).
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final_check; mkdir -p $O
cd $GRAFT_REPO_ROOT
flags() {
  case $1 in
    cfg2_mutated) echo "--workload cfg2 --dist mutated" ;; cfg3_mutated) echo "--workload cfg3 --dist mutated" ;; cfg4_mutated) echo "--workload cfg4 --dist mutated" ;;
    cfg2_ragged) echo "--workload cfg2 --dist ragged" ;; cfg2_dna) echo "--workload cfg2 --dist dna" ;; cfg2_dna5) echo "--workload cfg2 --dist dna5" ;;
    cfg2w_prefilter) echo "--workload cfg2w --unit-prefilter" ;; cfg2w_mutated) echo "--workload cfg2w --dist mutated" ;;
    cfg2w_mutated_prefilter) echo "--workload cfg2w --dist mutated --unit-prefilter" ;; cfg4w_prefilter) echo "--workload cfg4w --unit-prefilter" ;;
    cfg5w_220) echo "--workload cfg5w --costs 2,2,0,-" ;; cfg5w_231) echo "--workload cfg5w --costs 2,3,1,-" ;; cfg5w_2213) echo "--workload cfg5w --costs 2,2,1,3" ;;
    cfg5w_1101) echo "--workload cfg5w --costs 1,1,0,1" ;;
    hsearch8) echo "--workload hsearch --needle-len 8" ;; hsearch16) echo "--workload hsearch --needle-len 16" ;; hsearch32) echo "--workload hsearch --needle-len 32" ;;
    hsearch64) echo "--workload hsearch --needle-len 64" ;; *) echo "--workload $1" ;;
  esac
}
steps() { case $1 in cfg3) echo "--steps 3 --warmup 1" ;; cfg3_mutated*|cfg5*|hsearch*|cfg2t*) echo "--steps 10 --warmup 2" ;; *) echo "--steps 50" ;; esac; }
for tag in cfg2 cfg2_mutated cfg4 cfg4_mutated cfg1 cfg5 cfg3 cfg3_mutated cfg2w cfg4w cfg2l cfg2s cfg2t cfg2_ragged cfg2_dna cfg2_dna5 hsearch8 hsearch16 hsearch32 hsearch64 \
           cfg5w_220 cfg5w_231 cfg5w_2213 cfg5w_1101 cfg2w_prefilter cfg2w_mutated cfg2w_mutated_prefilter cfg4w_prefilter; do
  TA_NOENV=1 timeout 600 python bench.py $(flags $tag) $(steps $tag) --no-cpu --no-pmc > $O/bench_$tag.json 2> $O/bench_$tag.err
done
python - <<PY
import json, os
O = "$O"; P = "profiles/r05"
print("| workload | committed profile, ms | last build, ms | ratio |"); print("|---|---|---|---|")
for f in sorted(os.listdir(O)):
    if not f.endswith(".json"): continue
    try:
        new = json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])["ms_per_step"]
        old = json.loads(open(os.path.join(P, f)).read().strip().splitlines()[-1])["ms_per_step"]
        print("| %s | %.4f | %.4f | %.3f |" % (f[6:-5], old, new, new / old))
    except Exception as e:
        print("|", f, "| error:", e, "|")
PY
