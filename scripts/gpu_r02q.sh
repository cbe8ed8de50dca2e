#!/bin/bash
# session q: stride-8 window form: A/B against the static 33-bit form, parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
AB_WORKLOADS=cfg2 bash scripts/gpu_ab.sh 3 ab/onebit.so ab/s8.so | tee $O/ab_s8.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "bits or batch or kats or exp" 2>&1 | tail -3 | tee $O/pytest.txt
timeout 400 env TA_TUNING=1 python scripts/fuzz.py 4 5151 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
