#!/bin/bash
# cfg2 / cfg4 bench lines with the anti-diagonal CPU baseline; host CPU identification.
mkdir -p gpurun_out/n; O=$GRAFT_REPO_ROOT/gpurun_out/n
cd $GRAFT_REPO_ROOT
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket|Flags" | cut -c1-400 > $O/lscpu.txt
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 900 python bench.py --workload cfg4 --steps 20 --warmup 3 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
for f in $O/bench_cfg*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], d["value"], d["ms_per_step"], json.dumps(d["cpu_baseline"]))
PY
done
head -3 $O/lscpu.txt | cut -c1-120; tail -3 $O/bench_cfg2.err
