"""Round-4 probe A (GPU box): RCCL with ONE rank, hamming_search rates, CSR chunk form vs fixed line form on equal lengths."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import triple_accel_amd as T
from triple_accel_amd import batch as B
from triple_accel_amd import dist as TD
import torch.distributed as dist

def t_ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    for _ in range(200): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

# ---- 1. RCCL, one rank
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29711")
try:
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    g = Dg.rng(5)
    needle = Dg.rand_str(g, 24)
    hay = Dg.planted_haystack(11, needle, 600_000, 5000, 6)
    sh = B.haystack_tensor(hay)
    for st in (T.SearchType.All, T.SearchType.Best):
        ms = TD.levenshtein_search_sharded(needle, sh, 8, st, T.EditCosts(1, 1, 0, None))
        print("nccl world 1 sharded search", st, len(ms), ms[:2])
    x = TD.all_gather_results(torch.arange(7, dtype=torch.int32, device="cuda"))
    print("nccl all_gather_results", x.tolist(), x.device)
    dist.destroy_process_group()
except Exception as e:
    print("NCCL world-1 FAILED:", type(e).__name__, e)

# ---- 2. CSR chunk form vs fixed line form, equal lengths
LEV = (1, 1, 0, None)
for L in (144, 256):
    n = 1_000_000
    a, b = Dg.pairs_random(7, n, L)
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    tf = t_ms(lambda: B.levenshtein_k_batch(sa, sb, 32, LEV, out=out))
    kf = T.last_kernel_name()
    off = torch.arange(n + 1, dtype=torch.int64, device="cuda") * L
    ca = B.Strings(sa.blob, off, max_len=L); cb = B.Strings(sb.blob, off, max_len=L)
    out2 = torch.empty(n, dtype=torch.int32, device="cuda")
    tc = t_ms(lambda: B.levenshtein_k_batch(ca, cb, 32, LEV, out=out2))
    kc = T.last_kernel_name()
    assert torch.equal(out, out2)
    print("L=%d fixed %.4f ms (%s) | CSR same data %.4f ms (%s)" % (L, tf, kf, tc, kc))

# ---- 3. hamming_search
g = Dg.rng(1)
hay_np = Dg.random_bytes(g, 1 << 30)
needle = Dg.random_bytes(g, 32).tobytes()
for pos in range(1 << 16, hay_np.size - 100, 1 << 20):
    hay_np[pos:pos + 32] = np.frombuffer(needle, dtype=np.uint8)
    hay_np[pos + 5] = 7
hay = B.haystack_tensor(hay_np)
for nlen in (8, 32, 128):
    nd = needle[:nlen] if nlen <= 32 else (needle * 4)[:nlen]
    ms = t_ms(lambda: B.hamming_search_dev(nd, hay, nlen // 4), reps=5)
    print("hamming_search needle %d over 1 GiB: %.3f ms -> %.0f GB/s (%.3f of 8 TB/s), kernel %s" % (nlen, ms, (1 << 30) / ms / 1e6, (1 << 30) / ms / 1e6 / 8000, T.last_kernel_name()))
