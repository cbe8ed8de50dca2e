#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s_cfg5b; mkdir -p $O
run() { python bench.py --workload cfg5 --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4), round(d['ms_per_step'],4), round(d['value']))"; }
for rep in 1 2; do
  echo "default: $(run)"
  for t in 1024 1536 3072; do echo "tile $t: $(TA_TUNING=1 TA_FILTER_TILE=$t run)"; done
  echo "done groups 1: $(TA_TUNING=1 TA_SRCH_DONE_GROUPS=1 run)"
  echo "grid 256: $(TA_TUNING=1 TA_SRCH_GRID=256 run)"
  echo "grid 1024: $(TA_TUNING=1 TA_SRCH_GRID=1024 run)"
  echo "grid 128: $(TA_TUNING=1 TA_SRCH_GRID=128 run)"
done 2>&1 | tee $O/ab.txt
for v in "" "TA_SRCH_DONE_GROUPS=1" "TA_SRCH_GRID=256" "TA_SRCH_GRID=128"; do
(cd /tmp; rm -rf /tmp/kt; env TA_TUNING=1 $v rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu >/dev/null 2>&1; echo "== $v"; grep -E "lev_search_wave_kernel<false, true>|lev_filter" $(find /tmp/kt -name "kt_kernel_stats.csv" | head -1) | cut -c1-60,150-260)
done 2>&1 | tee $O/trace.txt
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
