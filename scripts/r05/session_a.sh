#!/bin/bash
# round 5, session A: new parity tests (weighted search through the superset filter, CSR trace batch), the driver's own bench command
# with the live issue counters + side batch, cfg5w with and without the weighted filter, first timings of exp on similar strings.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_trace.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 600 python bench.py > $O/default_cmd.json 2> $O/default_cmd.err; cut -c1-400 $O/default_cmd.json
for c in "2,2,0,-" "2,3,1,-" "2,2,1,3" "1,1,0,1"; do
  t=$(echo $c | tr -d ',')
  timeout 300 python bench.py --workload cfg5w --costs "$c" --steps 10 --no-cpu --no-pmc > $O/cfg5w_$t.json 2> $O/cfg5w_$t.err
  TA_TUNING=1 TA_SEARCH_NOWFILTER=1 timeout 300 python bench.py --workload cfg5w --costs "$c" --steps 3 --no-cpu --no-pmc > $O/cfg5w_${t}_nofilter.json 2> $O/cfg5w_${t}_nofilter.err
  python - <<PY
import json
for f in ("$O/cfg5w_$t.json", "$O/cfg5w_${t}_nofilter.json"):
    try:
        d = json.load(open(f)); print(f, round(d["ms_per_step"], 4), d["roofline"]["kernel_name"])
    except Exception as e: print(f, "failed", e)
PY
done
timeout 300 python bench.py --workload cfg5 --steps 10 --no-cpu --no-pmc > $O/cfg5.json 2>/dev/null; python -c "import json;d=json.load(open('$O/cfg5.json'));print('cfg5', d['ms_per_step'])"
for w in cfg3 cfg4; do
  timeout 600 python bench.py --workload $w --dist mutated --steps 3 --no-cpu --no-pmc > $O/${w}_mutated.json 2> $O/${w}_mutated.err
  python -c "import json;d=json.load(open('$O/${w}_mutated.json'));print('$w mutated', d['ms_per_step'], d['value'], d['roofline']['kernel_name'])"
done
