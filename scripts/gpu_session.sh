#!/bin/bash
# One GPU session: `gpurun -- bash scripts/gpu_session.sh <name> <part> [<part> ...]`; outputs under gpurun_out/<name>/.
# parts: tests (pytest -m gpu) | bench (device ms of every workload, no CPU leg) | cfg5 (search pass piece by piece + trace + LDS counters)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
NAME=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$NAME; mkdir -p $O
run() { python bench.py "$@" --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4), round(d['ms_per_step'],4), round(d['value']), d['kernel'].get('kernel'))"; }
trace() {  # trace <tag> <bench args...>
  local tag=$1; shift
  (cd /tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu 2>/dev/null | grep '^{' > $O/${tag}_under_trace.json; cp $(find /tmp/kt -name "kt_kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv)
  head -8 $O/${tag}_kernel_stats.csv | cut -c1-200
}
for part in "$@"; do
case $part in
tests)
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt ;;
bench)
  for rep in 1 2; do
    for wl in cfg2 cfg4 cfg5 cfg1; do echo "$wl: $(run --workload $wl --steps 50 --warmup 5)"; done
  done 2>&1 | tee $O/bench.txt ;;
cfg5)
  python scripts/measure_search_parts.py 2>&1 | tee $O/search_parts.txt
  trace cfg5 --workload cfg5 --steps 10 --warmup 2
  python scripts/pmc_collect.py --out $O/cfg5_pmc.json --workload cfg5 --sets sq1,sq2 --steps 5 2>&1 | tail -2 ;;
ragged)
  timeout 1200 python -m pytest tests/test_gpu_lev_batch.py tests/test_gpu_edge.py tests/test_gpu_search.py -x -q -k "ragged or length_order or generic or naive_contract" 2>&1 | tail -5 | tee $O/pytest_ragged.txt
  for rep in 1 2; do
    for wl in cfg2 cfg4 cfg2w cfg4w; do
      echo "$wl fixed: $(run --workload $wl --steps 30 --warmup 5)"
      echo "$wl ragged, length-ordered: $(run --workload $wl --dist ragged --steps 30 --warmup 5)"
      echo "$wl ragged, batch order: $(TA_TUNING=1 TA_NO_LENGTH_ORDER=1 run --workload $wl --dist ragged --steps 30 --warmup 5)"
    done
  done 2>&1 | tee $O/ragged.txt
  python bench.py --workload cfg2 --dist ragged --steps 30 --no-cpu > $O/bench_cfg2_ragged.json 2>$O/bench_cfg2_ragged.err; cut -c1-600 $O/bench_cfg2_ragged.json; tail -3 $O/bench_cfg2_ragged.err ;;
band)
  for wl in cfg2w cfg4w; do
    python bench.py --workload $wl --steps 30 > $O/bench_$wl.json 2>$O/bench_$wl.err; cut -c1-400 $O/bench_$wl.json; tail -3 $O/bench_$wl.err
    trace $wl --workload $wl --steps 10 --warmup 2
    python scripts/pmc_collect.py --out $O/${wl}_pmc.json --workload $wl --sets sq1,sq2,fetch,write,rd_b --steps 5 2>&1 | tail -2
  done ;;
raggedtrace)
  trace cfg2_ragged --workload cfg2 --dist ragged --steps 10 --warmup 2
  trace cfg2w_ragged --workload cfg2w --dist ragged --steps 10 --warmup 2 ;;
newtests)
  timeout 1800 python -m pytest tests/test_gpu_lev_batch.py tests/test_gpu_edge.py tests/test_gpu_search.py tests/test_gpu_bench.py tests/test_gpu_threads.py -x -q 2>&1 | tail -8 | tee $O/pytest_new.txt ;;
persist)
  for rep in 1 2; do
    for g in "" 1024 2048 4096 8192; do
      echo "cfg2 persist=$g: $(TA_TUNING=1 TA_BITS_PERSIST=$g run --workload cfg2 --steps 50 --warmup 5)"
      echo "cfg4 persist=$g: $(TA_TUNING=1 TA_BITS_PERSIST=$g run --workload cfg4 --steps 50 --warmup 5)"
    done
    for l in "" 40000 32000; do
      echo "cfg2 ragged block_lds=$l: $(TA_TUNING=1 TA_BITS_BLOCK_LDS=$l run --workload cfg2 --dist ragged --steps 30 --warmup 5)"
    done
    echo "cfg1: $(run --workload cfg1 --steps 200 --warmup 20)"
  done 2>&1 | tee $O/persist.txt ;;
bits2)
  timeout 1800 python -m pytest tests/test_gpu_lev_bits.py tests/test_gpu_lev_batch.py -x -q -k "two_pairs or ragged or length_order" 2>&1 | tail -5 | tee $O/pytest_bits2.txt
  for rep in 1 2 3; do
    echo "cfg4 one pair per lane: $(run --workload cfg4 --steps 50 --warmup 5)"
    echo "cfg4 two pairs per lane: $(TA_TUNING=1 TA_BITS2=1 run --workload cfg4 --steps 50 --warmup 5)"
    echo "cfg4 two pairs per lane, 2 waves per block: $(TA_TUNING=1 TA_BITS2=1 TA_BITS_WPB=2 run --workload cfg4 --steps 50 --warmup 5)"
    echo "cfg4 two pairs per lane, 4 waves per block: $(TA_TUNING=1 TA_BITS2=1 TA_BITS_WPB=4 run --workload cfg4 --steps 50 --warmup 5)"
    echo "cfg2 ragged: $(run --workload cfg2 --dist ragged --steps 30 --warmup 5)"
  done 2>&1 | tee $O/bits2.txt
  TA_TUNING=1 TA_BITS2=1 trace cfg4_bits2 --workload cfg4 --steps 10 --warmup 2
  TA_TUNING=1 TA_BITS2=1 python scripts/pmc_collect.py --out $O/cfg4_bits2_pmc.json --workload cfg4 --sets sq1,sq2,rd_b,write --steps 5 2>&1 | tail -2 ;;
bandab)
  timeout 1800 python -m pytest tests/test_gpu_lev_batch.py tests/test_gpu_trace.py tests/test_gpu_kats.py -x -q 2>&1 | tail -5 | tee $O/pytest_band.txt
  for rep in 1 2; do
    for wl in cfg2w cfg4w; do echo "$wl: $(run --workload $wl --steps 30 --warmup 5)"; done
  done 2>&1 | tee $O/bandab.txt ;;
scoreab)   # round 3: the band kernel's score form (cells as gc (i+j) - dp) against its cost form
  timeout 1800 python -m pytest tests/test_gpu_lev_batch.py tests/test_gpu_trace.py tests/test_gpu_kats.py tests/test_gpu_edge.py -x -q 2>&1 | tail -5 | tee $O/pytest_band.txt
  for rep in 1 2; do
    for wl in cfg2w cfg4w; do
      echo "$wl score form: $(run --workload $wl --steps 30 --warmup 5)"
      echo "$wl cost form:  $(TA_TUNING=1 TA_NO_SCORE_FORM=1 run --workload $wl --steps 30 --warmup 5)"
    done
    echo "cfg2 (unit costs through the DP kernel) score form: $(TA_TUNING=1 TA_NO_BITS=1 run --workload cfg2 --steps 20 --warmup 3)"
    echo "cfg2 (unit costs through the DP kernel) cost form:  $(TA_TUNING=1 TA_NO_BITS=1 TA_NO_SCORE_FORM=1 run --workload cfg2 --steps 20 --warmup 3)"
    echo "cfg2w ragged score form: $(run --workload cfg2w --dist ragged --steps 20 --warmup 3)"
  done 2>&1 | tee $O/scoreab.txt ;;
sizes)
  for rep in 1 2; do
    for n in 500000 1000000 2000000 4000000; do
      echo "cfg4 n=$n one pair: $(TA_TUNING=1 TA_NO_BITS2=1 run --workload cfg4 --pairs $n --steps 30 --warmup 5)"
      echo "cfg4 n=$n two pairs: $(run --workload cfg4 --pairs $n --steps 30 --warmup 5)"
    done
    for n in 500000 1000000 2000000; do echo "cfg2 n=$n: $(run --workload cfg2 --pairs $n --steps 30 --warmup 5)"; done
  done 2>&1 | tee $O/sizes.txt ;;
early)
  timeout 1200 python -m pytest tests/test_gpu_lev_bits.py -x -q -k "early_out or two_pairs or cfg" 2>&1 | tail -4 | tee $O/pytest_early.txt
  for rep in 1 2; do
    for wl in cfg2 cfg4; do
      echo "$wl random: $(run --workload $wl --steps 50 --warmup 5)"
      echo "$wl random, early out: $(run --workload $wl --steps 50 --warmup 5 --early-out)"
      echo "$wl mutated, early out: $(run --workload $wl --dist mutated --steps 50 --warmup 5 --early-out)"
    done
    echo "cfg2 geometry through the DP band kernel: $(TA_TUNING=1 TA_NO_BITS=1 run --workload cfg2 --steps 20 --warmup 3)"
    echo "cfg4 geometry through the DP band kernel: $(TA_TUNING=1 TA_NO_BITS=1 run --workload cfg4 --steps 20 --warmup 3)"
  done 2>&1 | tee $O/early.txt ;;
dna)
  timeout 1200 python -m pytest tests/test_gpu_lev_bits.py -x -q -k "small_alphabet" 2>&1 | tail -6 | tee $O/pytest_dna.txt
  for rep in 1 2; do
    for wl in cfg2 cfg4; do
      echo "$wl random bytes: $(run --workload $wl --steps 50 --warmup 5)"
      echo "$wl dna, small-alphabet kernel: $(run --workload $wl --dist dna --steps 50 --warmup 5)"
      echo "$wl dna, byte-test kernel: $(TA_TUNING=1 TA_NO_BITSQ=1 run --workload $wl --dist dna --steps 50 --warmup 5)"
    done
  done 2>&1 | tee $O/dna.txt
  trace cfg2_dna --workload cfg2 --dist dna --steps 10 --warmup 2
  python scripts/pmc_collect.py --out $O/cfg2_dna_pmc.json --workload cfg2 --sets sq1,sq2,rd_b,write --steps 5 --extra "--dist dna" 2>&1 | tail -2 ;;
benchlines)   # the bench lines again, now that profiles/<round>/ holds this build's counter passes (roofline.traffic, valu_issue)
  mkdir -p $O/lines
  for tag in cfg2 cfg2_mutated cfg4 cfg1 cfg5 cfg3 cfg2w cfg4w cfg2_ragged cfg2_dna; do
    case $tag in cfg2_mutated) fl="--workload cfg2 --dist mutated" ;; cfg2_ragged) fl="--workload cfg2 --dist ragged" ;; cfg2_dna) fl="--workload cfg2 --dist dna" ;; *) fl="--workload $tag" ;; esac
    case $tag in cfg3) st="--steps 3 --warmup 1" ;; cfg5) st="--steps 10 --warmup 2" ;; cfg2) st="" ;; *) st="--steps 50" ;; esac
    nocpu="--no-cpu"; [ $tag = cfg2 ] && nocpu=""
    timeout 900 python bench.py $fl $st $nocpu > $O/lines/bench_$tag.json 2> $O/lines/bench_$tag.err
    cut -c1-150 $O/lines/bench_$tag.json; grep -h "not spliced" $O/lines/bench_$tag.err
  done
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/lines/bench_driver_command.json 2>/dev/null; cut -c1-200 $O/lines/bench_driver_command.json ;;
fuzz)
  timeout 900 python scripts/fuzz.py 8 20260929 2>&1 | tail -5 | tee $O/fuzz.txt ;;
cfg5ab)
  bash scripts/gpu_session_cfg5.sh 2>&1 | tail -40; bash scripts/gpu_session_cfg5b.sh 2>&1 | tail -40 ;;
*) echo "unknown part $part" ;;
esac
done
