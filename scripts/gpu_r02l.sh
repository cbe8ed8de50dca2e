#!/bin/bash
# session l: count on the window's top diagonal (one v_alignbit per column) + `a` stored XOR 0x0C in the sliding form: A/B against the
# previous build, then the parity suites of the band kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02l
bash scripts/gpu_ab.sh 3 ab/base.so ab/top.so > gpurun_out/r02l/ab_top_diagonal.txt 2>&1
cat gpurun_out/r02l/ab_top_diagonal.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "bits or batch or kats or exp or threads or bench" 2>&1 | tail -5 | tee gpurun_out/r02l/pytest.txt
