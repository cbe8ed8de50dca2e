export TA_TUNING=1 TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "" "TA_TRACE_SKIP_WALK=1"; do
  (cd /tmp; rm -rf /tmp/kt; env $v rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2t --steps 3 --warmup 1 --no-cpu --no-pmc > /dev/null 2>&1; echo "== $v"; grep -E "trace_kernel|ckpt_kernel" $(find /tmp/kt -name "kt_kernel_stats.csv" | head -1) | cut -d, -f1-4 | cut -c1-140)
done
