cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s13
timeout 900 python -m pytest tests/test_gpu_edge.py -x -q -k "queue" 2>&1 | grep -v "^  File\|^Extension" | tail -25 | tee gpurun_out/s13/pytest_queue.txt
timeout 900 python -m pytest tests/test_gpu_lev_bits.py -x -q -k "small_alphabet" 2>&1 | grep -v "^  File\|^Extension" | tail -25 | tee gpurun_out/s13/pytest_dna.txt
AB_WORKLOADS="cfg2" AB_FLAGS="--dist dna" bash scripts/gpu_ab.sh 3 ab/bitsq_v1.so ab/bitsq_v2.so 2>&1 | tee gpurun_out/s13/ab_bitsq.txt
