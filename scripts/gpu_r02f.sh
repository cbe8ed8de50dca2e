#!/bin/bash
# round 2, GPU session f: single-pair kernel (tests + latency), full suite
export TMPDIR=/tmp TA_TUNING=1
O=$GRAFT_REPO_ROOT/gpurun_out/r02f; mkdir -p $O
cd $GRAFT_REPO_ROOT
python scripts/measure_latency.py > $O/latency.txt 2>&1; cat $O/latency.txt
TA_NO_ONE=1 python scripts/measure_latency.py > $O/latency_no_one.txt 2>&1; cat $O/latency_no_one.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 600 python scripts/fuzz.py 4 20260928 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
