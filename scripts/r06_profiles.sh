#!/bin/bash
# Round 6's measurement pass on ONE GPU box (the successor of scripts/gpu_profiles.sh): every number quoted in README.md / DESIGN.md comes
# from the files this writes under gpurun_out/profiles/ (copied into profiles/r06/ afterwards; scripts/make_tables.py r06 tabulates them).
# Per tag: bench_<tag>.json (the bench line), bench_<tag>_kernel_stats.csv + bench_<tag>_under_kernel_trace.json (rocprofv3 --kernel-trace
# --stats of the same command), bench_<tag>_pmc.json (SQ counters + L2 fabric-side requests, separate passes).
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $O
cd $GRAFT_REPO_ROOT
(lscpu | head -25; echo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; rocminfo | grep -E "Marketing|Compute Unit|Max Clock|gfx" | head -12) > $O/host_cpu.txt 2>&1
flags() {
  case $1 in
    cfg2_mutated) echo "--workload cfg2 --dist mutated" ;;
    cfg3_mutated) echo "--workload cfg3 --dist mutated" ;;
    cfg4_mutated) echo "--workload cfg4 --dist mutated" ;;
    cfg2_ragged) echo "--workload cfg2 --dist ragged" ;;
    cfg2_dna) echo "--workload cfg2 --dist dna" ;;
    cfg2_dna5) echo "--workload cfg2 --dist dna5" ;;
    cfg2t_220) echo "--workload cfg2t --tcosts 2,2,0,-" ;;
    cfg2tp_220) echo "--workload cfg2tp --tcosts 2,2,0,-" ;;
    cfg2t_231) echo "--workload cfg2t --tcosts 2,3,1,-" ;;
    cfg2t_2213) echo "--workload cfg2t --tcosts 2,2,1,3" ;;
    cfg2t_dp) echo "--workload cfg2t" ;;
    cfg2tp_stile128) echo "--workload cfg2tp" ;;
    cfg5w_231) echo "--workload cfg5w --costs 2,3,1,-" ;;
    hsearch8) echo "--workload hsearch --needle-len 8" ;;
    hsearch16) echo "--workload hsearch --needle-len 16" ;;
    hsearch32) echo "--workload hsearch --needle-len 32" ;;
    hsearch64) echo "--workload hsearch --needle-len 64" ;;
    *) echo "--workload $1" ;;
  esac
}
steps() { case $1 in cfg3) echo "--steps 3 --warmup 1" ;; cfg3_mutated) echo "--steps 10 --warmup 2" ;; cfg5*|hsearch*|cfg2t*) echo "--steps 10 --warmup 2" ;; cfg2) echo "" ;; *) echo "--steps 50" ;; esac; }
envof() { case $1 in cfg2t_dp) echo "TA_TUNING=1 TA_TRACE_NO_BITS=1" ;; cfg2tp_stile128) echo "TA_TUNING=1 TA_TRACE_STILE=128" ;; *) echo "TA_NOENV=1" ;; esac; }
TAGS="cfg2 cfg2_mutated cfg4 cfg4_mutated cfg1 cfg5 cfg3 cfg3_mutated cfg2w cfg4w cfg2l cfg2s cfg2t cfg2tp cfg2t_220 cfg2tp_220 cfg2t_231 cfg2t_2213 cfg2t_dp cfg2tp_stile128 cfg2_ragged cfg2_dna cfg2_dna5 cfg5w_231 hsearch8 hsearch16 hsearch32 hsearch64"
[ -n "$R06_TAGS" ] && TAGS="$R06_TAGS"
for tag in $TAGS; do
  extra="--no-cpu --no-pmc --no-all-configs"; [ $tag = cfg2 ] && extra=""       # (cfg2: the driver's command -- cpu_baseline, live counters, all_configs)
  env $(envof $tag) timeout 1500 python bench.py $(flags $tag) $(steps $tag) $extra > $O/bench_$tag.json 2> $O/bench_$tag.err
done
for tag in $TAGS; do
  [ $tag = cfg2_mutated ] && continue
  st=5; [ $tag = cfg3 ] && st=3
  (cd /tmp; rm -rf /tmp/kt_$tag; env $(envof $tag) rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py $(flags $tag) --steps $st --warmup 1 --no-cpu --no-pmc --no-all-configs 2>/dev/null | grep '^{' > $O/bench_${tag}_under_kernel_trace.json; cp $(find /tmp/kt_$tag -name "kt_kernel_stats.csv" | head -1) $O/bench_${tag}_kernel_stats.csv; rm -rf /tmp/kt_$tag)
  wl=$(flags $tag | cut -d' ' -f2); extra=$(flags $tag | cut -s -d' ' -f3-)
  env $(envof $tag) python scripts/pmc_collect.py --out $O/bench_${tag}_pmc.json --workload $wl --sets sq1,sq2,fetch,write,rd_b --steps $st --extra "$extra" 2>&1 | tail -1
done
# the device set (one process; this box has one GPU: device 0 listed N times) and the weak / strong figures of its bench mode
for n in 1 2 8; do
  timeout 600 python bench.py --single-process --gpus $n --steps 20 > $O/bench_sp_cfg2_weak_n$n.json 2> $O/bench_sp_cfg2_weak_n$n.err
  timeout 600 python bench.py --single-process --gpus $n --scaling strong --steps 20 > $O/bench_sp_cfg2_strong_n$n.json 2> $O/bench_sp_cfg2_strong_n$n.err
done
timeout 900 python bench.py --single-process --gpus 8 --workload cfg5 --pairs 128 --steps 10 > $O/bench_sp_cfg5_n8.json 2> $O/bench_sp_cfg5_n8.err
timeout 900 python bench.py --single-process --gpus 1 --workload cfg5 --pairs 1024 --steps 10 > $O/bench_sp_cfg5_n1.json 2> $O/bench_sp_cfg5_n1.err
python scripts/measure_latency.py > $O/latency.txt 2>&1
./scripts/ubench_mix > $O/ubench_mix.txt 2>&1
for f in $O/bench_*.json; do echo $f; cut -c1-160 $f; done
