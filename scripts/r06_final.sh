#!/bin/bash
# round 6, last session on the final build: the whole GPU suite, smoke(), the driver's default bench line, the device-set bench lines, a fuzz record
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $O
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/gpu_suite.txt 2>&1; cat $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
( time python bench.py ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -4 $O/bench_cfg2.err
for n in 1 2 8; do
  python bench.py --single-process --gpus $n --steps 20 > $O/bench_sp_cfg2_weak_n$n.json 2>/dev/null
  python bench.py --single-process --gpus $n --scaling strong --steps 20 > $O/bench_sp_cfg2_strong_n$n.json 2>/dev/null
done
python bench.py --single-process --gpus 8 --workload cfg5 --pairs 128 --steps 10 > $O/bench_sp_cfg5_n8.json 2>/dev/null
python bench.py --single-process --gpus 1 --workload cfg5 --pairs 1024 --steps 10 > $O/bench_sp_cfg5_n1.json 2>/dev/null
FUZZ_BACKTRACE=1 timeout 500 python scripts/r06/fuzz_r06.py 6 99991 2>&1 | tail -3 > $O/fuzz_final.txt; cat $O/fuzz_final.txt
cut -c1-200 $O/bench_cfg2.json
