#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_search.py -x -q 2>&1 | tail -2
python - <<PY
import os, sys, time, ctypes as C
import numpy as np, torch
ROOT=os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+"/tests")
import datagen as Dg
from triple_accel_amd import batch as B, _native as N
needle = Dg.random_bytes(Dg.rng(0x7A05), 32).tobytes()
hay_np = Dg.random_bytes(Dg.rng(1), 1 << 30)
t, length = B.haystack_tensor(hay_np)
cap = 1 << 22; hb = B._hit_buffer(t.device, cap); cnt = C.c_uint64(); cc = B._costs((1, 1, 0, None))._c()
for ft in (1024, 2048, 4096, 8192, 16384):
    os.environ["TA_FILTER_TILE"] = str(ft)
    N.lib().ta_levenshtein_search_dev(needle, 32, t.data_ptr(), length, 16, C.byref(cc), 0, 0, 0, hb.data_ptr(), cap, C.byref(cnt), None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        N.lib().ta_levenshtein_search_dev(needle, 32, t.data_ptr(), length, 16, C.byref(cc), 0, 0, 0, hb.data_ptr(), cap, C.byref(cnt), None)
    print("filter tile %d: C call %.3f ms" % (ft, (time.perf_counter() - t0) * 1e3 / 5), flush=True)
PY
