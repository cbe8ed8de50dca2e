#!/bin/bash
# session n: top-diagonal count, final form -- full GPU suite, fuzz, and the mutated (ragged) batch against the previous build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
timeout 600 env TA_TUNING=1 python scripts/fuzz.py 6 4242 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
cp triple_accel_amd/libtriple_accel_amd.so /tmp/ta_keep.so
for rep in 1 2; do for so in base top2; do
  cp ab/$so.so triple_accel_amd/libtriple_accel_amd.so
  for args in "--workload cfg2 --dist mutated" "--workload cfg2" "--workload cfg4"; do
    echo "$so $args $(python bench.py $args --steps 100 --warmup 10 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4), round(d['ms_per_step'],4))")"
  done
done; done | tee $O/ab_top2.txt
cp /tmp/ta_keep.so triple_accel_amd/libtriple_accel_amd.so
