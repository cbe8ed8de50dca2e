#!/bin/bash
# round 6, third GPU session: the whole GPU suite, then hamming_search routes, cfg3 on similar strings, the default bench line
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/gpu_suite.log; cat $O/gpu_suite.log
B="--steps 10 --warmup 2 --no-cpu --no-pmc --no-all-configs"
for n in 8 16 32 64; do
  timeout 600 python bench.py --workload hsearch --needle-len $n $B > $O/bench_hsearch$n.json 2> $O/bench_hsearch$n.err
done
timeout 600 python bench.py --workload cfg3 --dist mutated --steps 5 --warmup 1 --no-cpu --no-pmc --no-all-configs > $O/bench_cfg3_mutated.json 2> $O/bench_cfg3_mutated.err
timeout 600 python bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu --no-pmc --no-all-configs > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 600 python bench.py --no-cpu --no-pmc --no-all-configs > $O/bench_cfg2_overlap.json 2> $O/bench_cfg2_overlap.err
timeout 600 python bench.py --workload cfg4 --no-cpu --no-pmc --no-all-configs > $O/bench_cfg4_overlap.json 2> $O/bench_cfg4_overlap.err
for f in hsearch8 hsearch16 hsearch32 hsearch64 cfg3_mutated cfg3 cfg2_overlap cfg4_overlap; do python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_$f.json") if l.startswith("{")][-1])
    print("$f", round(d["ms_per_step"],4), round(d["value"]), d["roofline"]["kernel_name"], round(d["roofline"]["frac"],4), d.get("overlapped_passes"))
except Exception as e:
    print("$f", "FAILED", e); print(open("$O/bench_$f.err").read()[-600:])
PY
done
