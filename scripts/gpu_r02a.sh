#!/bin/bash
# round 2, GPU session a: counter list, A/B of the band-kernel loop variants, occupancy sweep, the new tests, the bench line
export TMPDIR=/tmp TA_TUNING=1
O=$GRAFT_REPO_ROOT/gpurun_out/r02a; mkdir -p $O
cd $GRAFT_REPO_ROOT
(rocprofv3 -L 2>&1 || rocprofv3 --list-avail 2>&1) > $O/counters_all.txt
grep -i -E "TCC_EA|TCC_REQ|TCC_HIT|TCC_MISS|FETCH|WRITE_SIZE|TCP_TCC|SQ_INSTS_VALU|SQ_ACTIVE_INST|SQ_WAIT|SQ_INST_CYCLES|SQ_BUSY|VALU" $O/counters_all.txt | cut -c1-220 | sort -u | head -150 > $O/counters_grep.txt
bash scripts/gpu_ab.sh 3 ab/r01.so ab/sl_pack0.so ab/sl_pack1.so > $O/ab.txt 2>&1
for lds in 0 40000 53000 80000; do
  t=$(TA_BITS_BLOCK_LDS=$lds python bench.py --steps 100 --warmup 10 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_pass'],4), d['kernel']['lds_bytes'])")
  echo "block_lds=$lds -> $t" >> $O/occupancy.txt
done
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; cut -c1-600 $O/bench_cfg2.json
cat $O/ab.txt $O/occupancy.txt
