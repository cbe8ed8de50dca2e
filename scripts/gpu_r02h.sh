#!/bin/bash
# round 2, GPU session h: answer-word specialisation A/B, the driver's own bench command with and without the clock ramp, bits tests
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02h; mkdir -p $O
cd $GRAFT_REPO_ROOT
AB_WORKLOADS="cfg2" bash scripts/gpu_ab.sh 3 ab/prev.so ab/answ.so > $O/ab_answ.txt 2>&1; cat $O/ab_answ.txt
for pw in 0 300 0 300; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --prewarm-ms $pw 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prewarm', d['prewarm_ms'], 'ms_per_step', round(d['ms_per_step'],4), 'value', round(d['value']))" | tee -a $O/prewarm.txt
done
timeout 900 python -m pytest tests/test_gpu_lev_bits.py tests/test_gpu_lev_batch.py tests/test_gpu_bench.py -x -q 2>&1 | tail -2
