#!/bin/bash
# session r: stride-8 form with one loop per block kind and the SDWA byte compare: A/B, parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
AB_WORKLOADS=cfg2 bash scripts/gpu_ab.sh 3 ab/s8.so ab/s8b.so | tee $O/ab_s8b.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "bits or batch or kats or exp" 2>&1 | tail -3 | tee $O/pytest.txt
timeout 300 env TA_TUNING=1 python scripts/fuzz.py 3 6161 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
