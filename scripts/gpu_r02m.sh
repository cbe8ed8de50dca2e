#!/bin/bash
# session m: why cfg2's time fell by 3.3 % when its instruction count fell by 7 %: SQ counters of both builds, and the no-HBM bound
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02m
cp triple_accel_amd/libtriple_accel_amd.so /tmp/ta_keep.so
for so in base top; do
  cp ab/$so.so triple_accel_amd/libtriple_accel_amd.so
  python scripts/pmc_collect.py --out gpurun_out/r02m/pmc_cfg2_$so.json --workload cfg2 --sets sq1,sq2 --steps 5 2>&1 | tail -1
  python scripts/exp_memory_bound.py > gpurun_out/r02m/membound_cfg2_$so.txt 2>&1
  cat gpurun_out/r02m/membound_cfg2_$so.txt
done
cp /tmp/ta_keep.so triple_accel_amd/libtriple_accel_amd.so
python - <<'PY'
import json
for so in ("base","top"):
    p=json.load(open("gpurun_out/r02m/pmc_cfg2_%s.json"%so))
    print(so, {k:(round(v["mean_per_launch"]) if isinstance(v,dict) and "mean_per_launch" in v else None) for k,v in p.items() if k.startswith(("SQ_","GRBM"))})
PY
