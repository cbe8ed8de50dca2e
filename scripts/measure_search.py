"""Host-side time breakdown of one cfg5-shaped search call (C ABI call vs Python post-processing)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import ctypes as C
from triple_accel_amd import batch as B, dist as TD, _native as N
g = Dg.rng(1)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
needle = Dg.random_bytes(Dg.rng(0x7A05), 32).tobytes()
hay_np = Dg.random_bytes(g, mib << 20)
for pos in range(1 << 16, hay_np.size - 100, 1 << 20):
    mm = np.frombuffer(Dg.mutate(g, needle, 10), dtype=np.uint8); hay_np[pos:pos + mm.size] = mm
hay = B.haystack_tensor(hay_np)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hits = B.levenshtein_search_dev(needle, hay, 16)
    t1 = time.perf_counter()
    best = TD.fold_best(hits, 16, True)
    t2 = time.perf_counter()
    print("search_dev %.3f ms (hits %d)  fold_best %.3f ms" % ((t1 - t0) * 1e3, len(hits), (t2 - t1) * 1e3), flush=True)
# raw C call only
t, length = hay
cap = 1 << 22
hb = B._hit_buffer(t.device, cap)
cnt = C.c_uint64(); cc = B._costs((1, 1, 0, None))._c()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = N.lib().ta_levenshtein_search_dev(needle, 32, t.data_ptr(), length, 16, C.byref(cc), 0, 0, 0, hb.data_ptr(), cap, C.byref(cnt), None)
    t1 = time.perf_counter()
    print("C call %.3f ms rc=%d count=%d" % ((t1 - t0) * 1e3, rc, cnt.value), flush=True)
for ft in (512, 1024, 4096, 8192):
    os.environ["TA_FILTER_TILE"] = str(ft)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        N.lib().ta_levenshtein_search_dev(needle, 32, t.data_ptr(), length, 16, C.byref(cc), 0, 0, 0, hb.data_ptr(), cap, C.byref(cnt), None)
    print("filter tile %d: C call %.3f ms" % (ft, (time.perf_counter() - t0) * 1e3 / 5), flush=True)
