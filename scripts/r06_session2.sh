#!/bin/bash
# round 6, second GPU session: the new tests, then the trace workloads (16-byte records, packed records, unit costs x 2, weighted costs)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_adversarial.py tests/test_gpu_trace.py -q 2>&1 | tail -15
B="--steps 10 --warmup 2 --no-cpu --no-pmc --no-all-configs"
timeout 600 python bench.py --workload cfg2t $B > $O/bench_cfg2t.json 2> $O/bench_cfg2t.err
timeout 600 python bench.py --workload cfg2tp $B > $O/bench_cfg2tp.json 2> $O/bench_cfg2tp.err
timeout 600 python bench.py --workload cfg2t --tcosts 2,2,0,- --tk 64 $B > $O/bench_cfg2t_220.json 2> $O/bench_cfg2t_220.err
timeout 600 python bench.py --workload cfg2tp --tcosts 2,2,0,- --tk 64 $B > $O/bench_cfg2tp_220.json 2> $O/bench_cfg2tp_220.err
timeout 600 python bench.py --workload cfg2t --tcosts 2,3,1,- --tk 64 $B > $O/bench_cfg2t_231.json 2> $O/bench_cfg2t_231.err
for f in cfg2t cfg2tp cfg2t_220 cfg2tp_220 cfg2t_231; do python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_$f.json") if l.startswith("{")][-1])
    print("$f", round(d["ms_per_step"],4), round(d["value"]), d["roofline"]["kernel_name"], d["roofline"]["algorithmic_bytes_per_pass"], round(d["roofline"]["frac"],4))
except Exception as e:
    print("$f", "FAILED", e); print(open("$O/bench_$f.err").read()[-600:])
PY
done
python scripts/pmc_collect.py --out $O/bench_cfg2tp_pmc.json --workload cfg2tp --sets sq1,rd_b,write,issue --steps 5 2>&1 | tail -2
