#!/bin/bash
# cfg5 session: search parity tests, then the pass timed under the A/B switches, kernel trace and LDS counters
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s_cfg5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_search.py tests/test_gpu_dist.py -x -q 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
run() { python bench.py --workload cfg5 --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4), round(d['ms_per_step'],4), round(d['value']))"; }
for rep in 1 2; do
  echo "default: $(run)"
  echo "flat table: $(TA_TUNING=1 TA_FILTER_FLAT=1 run)"
  for t in 2048 4096 16384; do echo "tile $t: $(TA_TUNING=1 TA_FILTER_TILE=$t run)"; done
  for t in 2048 4096 16384; do echo "flat tile $t: $(TA_TUNING=1 TA_FILTER_FLAT=1 TA_FILTER_TILE=$t run)"; done
done 2>&1 | tee $O/ab.txt
(cd /tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu 2>/dev/null | grep '^{' > $O/under_trace.json; cp $(find /tmp/kt -name "kt_kernel_stats.csv" | head -1) $O/kernel_stats.csv; cp $(find /tmp/kt -name "kt_hip_api_stats.csv" | head -1) $O/ 2>/dev/null)
head -12 $O/kernel_stats.csv
python scripts/pmc_collect.py --out $O/pmc.json --workload cfg5 --sets sq1,sq2 --steps 5 2>&1 | tail -2
