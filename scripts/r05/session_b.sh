#!/bin/bash
# round 5, session B: the phased bit-sliced hamming_search form -- parity on the device, then the bench rows for 8 / 16 / 32 / 64 / 128-byte needles
# with and without it (TA_HAMMING_SEARCH_NO_PHASE=1 = round 4's routing).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_search.py -x -q -m gpu -k "hamming" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for n in 8 16 32 64 128; do
  timeout 300 python bench.py --workload hsearch --needle-len $n --steps 10 --no-cpu --no-pmc > $O/hsearch$n.json 2> $O/hsearch$n.err
  TA_TUNING=1 TA_HAMMING_SEARCH_NO_PHASE=1 timeout 300 python bench.py --workload hsearch --needle-len $n --steps 5 --no-cpu --no-pmc > $O/hsearch${n}_nophase.json 2> $O/hsearch${n}_nophase.err
  python - <<PY
import json
for f in ("$O/hsearch$n.json", "$O/hsearch${n}_nophase.json"):
    try:
        d = json.load(open(f)); print(f, round(d["ms_per_step"], 4), d["roofline"]["kernel_name"], d["config"]["workload"][:60])
    except Exception as e: print(f, "failed", e)
PY
done
