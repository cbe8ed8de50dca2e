cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ragged_ab
run() { python bench.py --dist ragged --steps 40 --warmup 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4), round(d['value']))"; }
for rep in 1 2; do
  echo "default: $(run)"
  for w in 1 2; do echo "wpb $w: $(TA_TUNING=1 TA_BITS_WPB=$w run)"; done
  for l in 36000 30000 24000; do echo "block lds $l: $(TA_TUNING=1 TA_BITS_BLOCK_LDS=$l run)"; done
  echo "static window form (TA_BITS_STATIC=2): $(TA_TUNING=1 TA_BITS_STATIC=2 run)"
  echo "sliding window form (TA_BITS_STATIC=1): $(TA_TUNING=1 TA_BITS_STATIC=1 run)"
done 2>&1 | tee gpurun_out/ragged_ab/ab.txt
