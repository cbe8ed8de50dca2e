#!/bin/bash
# round-4 GPU session A: VLINE probe, kernel trace of the ragged bench, RCCL test, hsearch bench lines
mkdir -p gpurun_out/r04; O=gpurun_out/r04
python scripts/r04/probe_b.py > $O/probe_b3.txt 2>&1
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_rag -o rag -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2 --dist ragged --steps 50 --no-cpu > $GRAFT_REPO_ROOT/$O/bench_ragged_trace.json 2> $GRAFT_REPO_ROOT/$O/bench_ragged_trace.err)
cp $(find /tmp/prof_rag -name "*kernel_stats.csv" | head -1) $O/bench_cfg2_ragged_kernel_stats.csv
python -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -5 > $O/t_rccl.txt
for n in 8 32 128; do python bench.py --workload hsearch --needle-len $n --steps 10 > $O/bench_hsearch$n.json 2> $O/bench_hsearch$n.err; done
grep -E "^R|rror" $O/probe_b3.txt; head -8 $O/bench_cfg2_ragged_kernel_stats.csv | cut -c1-200; cat $O/t_rccl.txt; for n in 8 32 128; do cut -c1-330 $O/bench_hsearch$n.json; tail -2 $O/bench_hsearch$n.err; done
