#!/bin/bash
# round 6, first GPU session: the driver's default command (all five BASELINE configs in one line) + the single-process device-set bench
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
for n in 1 2 8; do
  timeout 600 python bench.py --single-process --gpus $n --pairs 250000 --steps 20 > $O/bench_sp_cfg2_n$n.json 2> $O/bench_sp_cfg2_n$n.err
  timeout 600 python bench.py --single-process --gpus $n --scaling strong --steps 20 > $O/bench_sp_cfg2_strong_n$n.json 2> $O/bench_sp_cfg2_strong_n$n.err
done
timeout 600 python bench.py --single-process --gpus 2 --workload cfg5 --pairs 512 --steps 10 > $O/bench_sp_cfg5_n2.json 2> $O/bench_sp_cfg5_n2.err
timeout 600 python bench.py --single-process --gpus 8 --workload cfg5 --pairs 128 --steps 10 > $O/bench_sp_cfg5_n8.json 2> $O/bench_sp_cfg5_n8.err
for f in $O/*.json; do echo $f; cut -c1-300 $f; done
tail -5 $O/*.err
