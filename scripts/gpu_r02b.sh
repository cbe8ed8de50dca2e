#!/bin/bash
# round 2, GPU session b: mixed-stream micro-benchmark, coalesced fetch A/B + size-resolved L2 request counters, new tests, latency
export TMPDIR=/tmp TA_TUNING=1
O=$GRAFT_REPO_ROOT/gpurun_out/r02b; mkdir -p $O
cd $GRAFT_REPO_ROOT
./scripts/ubench_mix > $O/ubench_mix.txt 2>&1; tail -30 $O/ubench_mix.txt
cp triple_accel_amd/libtriple_accel_amd.so ab/coop.so
bash scripts/gpu_ab.sh 3 ab/r01.so ab/coop.so > $O/ab_coop.txt 2>&1; cat $O/ab_coop.txt
for v in r01 coop; do
  cp ab/$v.so triple_accel_amd/libtriple_accel_amd.so
  python scripts/pmc_collect.py --out $O/traffic_cfg2_$v.json --workload cfg2 --sets tcc --steps 5 2>&1 | tail -2
  python scripts/pmc_collect.py --out $O/traffic_cfg2_${v}_run2.json --workload cfg2 --sets fetch,rd_b --steps 5 2>&1 | tail -1
done
cp ab/coop.so triple_accel_amd/libtriple_accel_amd.so
timeout 900 python -m pytest tests/test_gpu_threads.py tests/test_gpu_lev_bits.py tests/test_gpu_lev_batch.py tests/test_gpu_search.py tests/test_gpu_kats.py tests/test_gpu_trace.py -x -q > $O/pytest_new.txt 2>&1; tail -5 $O/pytest_new.txt
python scripts/measure_latency.py > $O/latency.txt 2>&1; cat $O/latency.txt
