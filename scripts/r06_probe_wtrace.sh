#!/bin/bash
# weighted tracebacks: where the time goes (trace kernel / walk kernel), k = 32 and 64
export TMPDIR=/tmp
cd /tmp
for tk in 32 64; do
  rm -rf /tmp/kt_w$tk
  rocprofv3 --kernel-trace --stats -d /tmp/kt_w$tk -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload cfg2t --tcosts 2,3,1,- --tk $tk --steps 5 --warmup 1 --no-cpu --no-pmc --no-all-configs --prewarm-ms 0 2>/dev/null | grep '^{' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("tk", '$tk', round(d["ms_per_step"],3), d["kernel"])'
  head -4 $(find /tmp/kt_w$tk -name "kt_kernel_stats.csv" | head -1) | cut -c1-200
done
