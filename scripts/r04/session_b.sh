#!/bin/bash
# round-4 GPU session B: VLINE (computed-jump bursts, branch-free wrap copies) -- tests, probe, counters for both fetch forms
mkdir -p gpurun_out/r04; O=gpurun_out/r04
python -m pytest tests/test_gpu_lev_bits.py tests/test_gpu_lev_batch.py -x -q 2>&1 | tail -5 > $O/t_b.txt
python scripts/r04/probe_b.py > $O/probe_b4.txt 2>&1
export TMPDIR=/tmp
python scripts/pmc_collect.py --out $O/bench_cfg2_ragged_pmc.json --workload cfg2 --extra "--dist ragged" > $O/pmc_vline.log 2>&1
TA_TUNING=1 TA_BITS_NO_VLINE=1 TA_ORDER_SHIFT3=1 python scripts/pmc_collect.py --out $O/bench_cfg2_ragged_chunk_pmc.json --workload cfg2 --extra "--dist ragged" > $O/pmc_chunk.log 2>&1
cat $O/t_b.txt; grep -E "^R|rror" $O/probe_b4.txt; tail -2 $O/pmc_vline.log; tail -2 $O/pmc_chunk.log
