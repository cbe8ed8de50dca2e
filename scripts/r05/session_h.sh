#!/bin/bash
# round 5, session H: the randomised parity run on the round's LAST build (folded forward sweep, pinned hamming report, resume entry, 512-byte filter)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles; mkdir -p $O
timeout 1700 python scripts/fuzz.py 26 777 > $O/fuzz_late.txt 2>&1; tail -2 $O/fuzz_late.txt
