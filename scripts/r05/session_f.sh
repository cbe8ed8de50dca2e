#!/bin/bash
# round 5, session F: the randomised parity run on the round's final build (scripts/fuzz.py with the round's paths: unit pre-pass option,
# device-driven exp rounds, checkpoint trace kernel with random tile sizes, phased hamming_search incl. four-letter texts), two seeds;
# the counter passes of the two pre-pass rows (their --extra needs the = form).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles; mkdir -p $O
timeout 900 python scripts/fuzz.py 12 20260930 > $O/fuzz_a.txt 2>&1; tail -2 $O/fuzz_a.txt
timeout 700 python scripts/fuzz.py 9 5151 > $O/fuzz_b.txt 2>&1; tail -2 $O/fuzz_b.txt
python scripts/pmc_collect.py --out $O/bench_cfg2w_prefilter_pmc.json --workload cfg2w --sets sq1,sq2,fetch,write,rd_b --steps 5 --extra=--unit-prefilter 2>&1 | tail -1 | cut -c1-200
