#!/bin/bash
# round 2, GPU session e: two pairs per lane (cfg4), cfg4 chunk-vs-line, bits tests
export TMPDIR=/tmp TA_TUNING=1
O=$GRAFT_REPO_ROOT/gpurun_out/r02e; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lev_bits.py tests/test_gpu_lev_batch.py tests/test_gpu_kats.py -x -q > $O/pytest_bits.txt 2>&1; tail -5 $O/pytest_bits.txt
for rep in 1 2 3; do
for env in "" TA_NO_BITS2=1; do
  t=$(env $env python bench.py --workload cfg4 --steps 100 --warmup 10 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4), d['kernel']['pairs_per_wave'], d['kernel']['lds_bytes'])")
  echo "cfg4 [$env] $t" | tee -a $O/ab_cfg4.txt
done; done
python bench.py --steps 100 --warmup 10 --no-cpu 2>/dev/null | cut -c1-250
