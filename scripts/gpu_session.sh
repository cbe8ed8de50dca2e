#!/bin/bash
# One GPU session: `gpurun -- bash scripts/gpu_session.sh <name> <part> [<part> ...]`; outputs under gpurun_out/<name>/.
# parts: tests (pytest -m gpu) | bench (device ms of every workload, no CPU leg) | cfg5 (search pass piece by piece + trace + LDS counters)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
NAME=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$NAME; mkdir -p $O
run() { python bench.py "$@" --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4), round(d['ms_per_step'],4), round(d['value']), d['kernel'].get('kernel'))"; }
trace() {  # trace <tag> <bench args...>
  local tag=$1; shift
  (cd /tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu 2>/dev/null | grep '^{' > $O/${tag}_under_trace.json; cp $(find /tmp/kt -name "kt_kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv)
  head -8 $O/${tag}_kernel_stats.csv | cut -c1-200
}
for part in "$@"; do
case $part in
tests)
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt ;;
bench)
  for rep in 1 2; do
    for wl in cfg2 cfg4 cfg5 cfg1; do echo "$wl: $(run --workload $wl --steps 50 --warmup 5)"; done
  done 2>&1 | tee $O/bench.txt ;;
cfg5)
  python scripts/measure_search_parts.py 2>&1 | tee $O/search_parts.txt
  trace cfg5 --workload cfg5 --steps 10 --warmup 2
  python scripts/pmc_collect.py --out $O/cfg5_pmc.json --workload cfg5 --sets sq1,sq2 --steps 5 2>&1 | tail -2 ;;
*) echo "unknown part $part" ;;
esac
done
