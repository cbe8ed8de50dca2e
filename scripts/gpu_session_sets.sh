# cfg2 device time against the number of resident sets (4,096 wavefronts = 262,144 pairs each): is the last, partial set a full round?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sets
run() { python bench.py "$@" --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4), round(d['ms_per_step'],4))"; }
for rep in 1 2; do
  for n in 131072 262144 393216 524288 655360 786432 917504 1000000 1048576 1179648 1310720 2097152; do
    echo "cfg2 pairs=$n sets=$(python -c "print(round($n/262144,2))"): $(run --workload cfg2 --pairs $n --steps 50 --warmup 5)"
  done
done 2>&1 | tee gpurun_out/sets/sets.txt
