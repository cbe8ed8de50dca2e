#!/bin/bash
# The round's measurement pass on ONE GPU box: every number quoted in README.md / DESIGN.md comes from the files this writes
# (copy gpurun_out/profiles/* into profiles/<round>/ afterwards, then run scripts/make_tables.py).
#   bench_cfg{1..5}.json          one bench line per BASELINE configuration (cfg2 with the cpu_baseline leg)
#   bench_cfg2_mutated.json       cfg2 on the "mutated" input distribution
#   bench_cfg{1..5}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command; bench_cfgN_under_kernel_trace.json = what
#                                 bench.py itself measured (HIP events) inside that profiled run
#   bench_cfg{1..5}_pmc.json      SQ counters (VALU instructions, busy cycles, waits, LDS) + L2 fabric-side requests, separate passes
#   host_cpu.txt, latency.txt, ubench_mix.txt, ubench_cellwidth.txt
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $O
cd $GRAFT_REPO_ROOT
(lscpu | head -25; echo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; rocminfo | grep -E "Marketing|Compute Unit|Max Clock|gfx" | head -12) > $O/host_cpu.txt 2>&1
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python bench.py --dist mutated --no-cpu > $O/bench_cfg2_mutated.json 2>/dev/null
timeout 600 python bench.py --workload cfg4 --steps 50 > $O/bench_cfg4.json 2>/dev/null
timeout 600 python bench.py --workload cfg1 --steps 50 > $O/bench_cfg1.json 2>/dev/null
timeout 900 python bench.py --workload cfg5 --steps 10 --warmup 2 > $O/bench_cfg5.json 2>/dev/null
timeout 900 python bench.py --workload cfg3 --steps 3 --warmup 1 > $O/bench_cfg3.json 2>/dev/null
for wl in cfg2 cfg4 cfg1 cfg5 cfg3; do
  steps=5; [ $wl = cfg3 ] && steps=3
  (cd /tmp; rm -rf /tmp/kt_$wl; rocprofv3 --kernel-trace --stats -d /tmp/kt_$wl -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps $steps --warmup 1 --no-cpu 2>/dev/null | grep '^{' > $O/bench_${wl}_under_kernel_trace.json; cp $(find /tmp/kt_$wl -name "kt_kernel_stats.csv" | head -1) $O/bench_${wl}_kernel_stats.csv; rm -rf /tmp/kt_$wl)
  python scripts/pmc_collect.py --out $O/bench_${wl}_pmc.json --workload $wl --sets sq1,sq2,fetch,write,rd_b --steps $steps 2>&1 | tail -1
done
python scripts/measure_latency.py > $O/latency.txt 2>&1
./scripts/ubench_mix > $O/ubench_mix.txt 2>&1
./scripts/ubench_cellwidth > $O/ubench_cellwidth.txt 2>&1
for f in $O/bench_cfg*.json; do echo $f; cut -c1-200 $f; done
