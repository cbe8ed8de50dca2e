#!/bin/bash
# round 2, GPU session g: final build -- full suite, smoke(), latency, a longer fuzz run, the default bench line, 2-rank bench lines
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02g; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python scripts/measure_latency.py > $O/latency.txt 2>&1; cat $O/latency.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
timeout 600 python bench.py --gpus 2 --no-cpu --steps 20 > $O/bench_gpus2_weak.json 2>/dev/null; cut -c1-200 $O/bench_gpus2_weak.json
timeout 900 env TA_TUNING=1 python scripts/fuzz.py 12 777 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
