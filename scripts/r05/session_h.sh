#!/bin/bash
# Round 5, last session: the whole GPU suite, smoke(), the trace route's randomised run, the general fuzz run, cfg2t's own-sweep A/B row again
# and the driver's default bench line -- on the round's last build.
export TMPDIR=/tmp
O=gpurun_out/session_h; mkdir -p $O gpurun_out/profiles
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python scripts/r05/fuzz_trace.py ${FUZZ_TRACE_MIN:-6} > $O/fuzz_trace.txt 2>&1; tail -8 $O/fuzz_trace.txt
timeout 900 python scripts/fuzz.py ${FUZZ_MIN:-6} > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
TA_TUNING=1 TA_TRACE_OWN_SWEEP=1 python bench.py --workload cfg2t --steps 10 --warmup 2 --no-cpu --no-pmc > gpurun_out/profiles/bench_cfg2t_own_sweep.json 2> gpurun_out/profiles/bench_cfg2t_own_sweep.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
python - <<PY
import json
for f in ("gpurun_out/profiles/bench_cfg2t_own_sweep.json",):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["ms_per_step"], 4), d["roofline"]["kernel_name"])
PY
