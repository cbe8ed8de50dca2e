#!/bin/bash
# session p: ONEBIT top word (the 33rd diagonal as match | carry): A/B and parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
AB_WORKLOADS=cfg2 bash scripts/gpu_ab.sh 3 ab/top2.so ab/onebit.so | tee $O/ab_onebit.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "bits or batch or kats or exp" 2>&1 | tail -3 | tee $O/pytest.txt
timeout 400 env TA_TUNING=1 python scripts/fuzz.py 4 9191 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
