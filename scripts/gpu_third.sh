#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
./scripts/ubench_valu > gpurun_out/ubench_valu.log 2>&1
cat gpurun_out/ubench_valu.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_kt -o kt -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_kt.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc1 -o pmc1 -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_pmc1.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_pmc1.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc2 -o pmc2 -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_pmc2.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_pmc2.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc3 -o pmc3 -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc4 -o pmc4 -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_pmc4.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out -name "*.csv" | head -30
