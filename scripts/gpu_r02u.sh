#!/bin/bash
# session u: final build -- full GPU suite, smoke(), the driver's own bench command, 2-rank lines, a longer fuzz run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; cut -c1-400 $O/bench_driver_cmd.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu 2>/dev/null | cut -c1-250
TA_BENCH_BACKEND=gloo python bench.py --gpus 2 --pairs 200000 --steps 10 --warmup 3 --no-cpu 2>/dev/null | cut -c1-600 | tee $O/bench_2ranks.json
timeout 800 env TA_TUNING=1 python scripts/fuzz.py 10 31337 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
