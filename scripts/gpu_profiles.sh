#!/bin/bash
# The round's measurement pass on ONE GPU box: every number quoted in README.md / DESIGN.md comes from the files this writes
# (copy gpurun_out/profiles/* into profiles/<round>/ afterwards, then run scripts/isa_mix.py and scripts/make_tables.py <round>).
# Workloads (tag = file stem): the five BASELINE configurations cfg1..cfg5 (cfg2 with the cpu_baseline leg), cfg2 on the mutated
# distribution, the general-cost geometries cfg2w / cfg4w (DP band-wavefront kernel), the ragged CSR batch cfg2_ragged
# (length-ordered on the device) and the DNA batch cfg2_dna (small-alphabet kernel).  Per tag:
#   bench_<tag>.json                  the bench line (driver protocol)
#   bench_<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command; bench_<tag>_under_kernel_trace.json = what
#                                     bench.py itself measured (HIP events) inside that profiled run
#   bench_<tag>_pmc.json              SQ counters (VALU instructions, busy cycles, waits, LDS) + L2 fabric-side requests, separate passes
# plus host_cpu.txt, latency.txt, ubench_mix.txt, ubench_cellwidth.txt, bench_cfg2_early_out.json, bench_cfg2_2m.json
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $O
cd $GRAFT_REPO_ROOT
(lscpu | head -25; echo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; rocminfo | grep -E "Marketing|Compute Unit|Max Clock|gfx" | head -12) > $O/host_cpu.txt 2>&1
flags() {   # tag -> bench.py flags
  case $1 in
    cfg2_mutated) echo "--workload cfg2 --dist mutated" ;;
    cfg2_ragged) echo "--workload cfg2 --dist ragged" ;;
    cfg2_dna) echo "--workload cfg2 --dist dna" ;;
    cfg2_dna5) echo "--workload cfg2 --dist dna5" ;;
    cfg2_protein_table) echo "--workload cfg2 --dist protein" ;;
    cfg2_ragged_vline) echo "--workload cfg2 --dist ragged" ;;
    hsearch8) echo "--workload hsearch --needle-len 8" ;;
    hsearch32) echo "--workload hsearch --needle-len 32" ;;
    hsearch64) echo "--workload hsearch --needle-len 64" ;;
    *) echo "--workload $1" ;;
  esac
}
steps() { case $1 in cfg3) echo "--steps 3 --warmup 1" ;; cfg5|hsearch*|cfg2t) echo "--steps 10 --warmup 2" ;; cfg2) echo "" ;; *) echo "--steps 50" ;; esac; }
# (cfg2_ragged_vline: the ragged batch through the VLINE fetch form, TA_TUNING=1 TA_BITS_VLINE=1 -- an A/B row, not a default path;
#  cfg2_protein_table: the 20 amino acids through the 5-bit-code small-alphabet kernel, TA_BITSQ_WIDE=1 -- an A/B row too: the default runs the byte test;
#  cfg2_dna5: A C G T N, where the default IS that kernel)
TAGS="cfg2 cfg2_mutated cfg4 cfg1 cfg5 cfg3 cfg2w cfg4w cfg2l cfg2s cfg2t cfg2_ragged cfg2_ragged_vline cfg2_dna cfg2_dna5 cfg2_protein_table hsearch8 hsearch32 hsearch64"
envof() { case $1 in cfg2_ragged_vline) echo "TA_TUNING=1 TA_BITS_VLINE=1" ;; cfg2_protein_table) echo "TA_TUNING=1 TA_BITSQ_WIDE=1" ;; *) echo "TA_NOENV=1" ;; esac; }
for tag in $TAGS; do
  nocpu="--no-cpu --no-pmc"; [ $tag = cfg2 ] && nocpu=""       # (cfg2: the driver's command -- cpu_baseline leg and the live counter passes included)
  env $(envof $tag) timeout 900 python bench.py $(flags $tag) $(steps $tag) $nocpu > $O/bench_$tag.json 2> $O/bench_$tag.err
done
timeout 600 python bench.py --early-out --no-cpu --no-pmc > $O/bench_cfg2_early_out.json 2>/dev/null
timeout 600 python bench.py --pairs 2000000 --no-cpu --no-pmc > $O/bench_cfg2_2m.json 2>/dev/null
for tag in $TAGS; do
  [ $tag = cfg2_mutated ] && continue
  st=5; [ $tag = cfg3 ] && st=3
  (cd /tmp; rm -rf /tmp/kt_$tag; env $(envof $tag) rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py $(flags $tag) --steps $st --warmup 1 --no-cpu --no-pmc 2>/dev/null | grep '^{' > $O/bench_${tag}_under_kernel_trace.json; cp $(find /tmp/kt_$tag -name "kt_kernel_stats.csv" | head -1) $O/bench_${tag}_kernel_stats.csv; rm -rf /tmp/kt_$tag)
  wl=$(flags $tag | cut -d' ' -f2); extra=$(flags $tag | cut -s -d' ' -f3-)
  env $(envof $tag) python scripts/pmc_collect.py --out $O/bench_${tag}_pmc.json --workload $wl --sets sq1,sq2,fetch,write,rd_b --steps $st --extra "$extra" 2>&1 | tail -1
done
python scripts/measure_latency.py > $O/latency.txt 2>&1
python scripts/measure_search_parts.py > $O/search_parts.txt 2>&1
./scripts/ubench_mix > $O/ubench_mix.txt 2>&1
./scripts/ubench_cellwidth > $O/ubench_cellwidth.txt 2>&1
./scripts/ubench_mix2 > $O/ubench_mix2.txt 2>&1
python scripts/r04/probe_b.py > $O/probe_ragged.txt 2>&1
python scripts/r04/probe_h.py > $O/probe_hsearch.txt 2>&1
for f in $O/bench_cfg*.json; do echo $f; cut -c1-160 $f; done
