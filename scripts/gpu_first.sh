#!/bin/bash
# First on-hardware pass: smoke, GPU parity tests, a short bench, kernel-trace profile.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > gpurun_out/rocminfo.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q -k "not search" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/bench.log
