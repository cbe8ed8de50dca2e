#!/bin/bash
# One workload of scripts/gpu_profiles.sh again (bench line, kernel trace, counter passes): gpu_profile_one.sh <tag> <bench flags...>
export TMPDIR=/tmp
tag=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py "$@" --steps 50 --no-cpu --no-pmc > $O/bench_$tag.json 2>/dev/null
(cd /tmp; rm -rf /tmp/kt_$tag; rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 5 --warmup 1 --no-cpu --no-pmc 2>/dev/null | grep '^{' > $O/bench_${tag}_under_kernel_trace.json; cp $(find /tmp/kt_$tag -name "kt_kernel_stats.csv" | head -1) $O/bench_${tag}_kernel_stats.csv; rm -rf /tmp/kt_$tag)
wl=$(echo "$@" | sed -n 's/.*--workload \([a-z0-9]*\).*/\1/p'); [ -z "$wl" ] && wl=cfg2
extra=$(echo "$@" | sed 's/--workload [a-z0-9]*//')
python scripts/pmc_collect.py --out $O/bench_${tag}_pmc.json --workload $wl --sets sq1,sq2,fetch,write,rd_b --steps 5 --extra "$extra" 2>&1 | tail -1
cut -c1-200 $O/bench_$tag.json
