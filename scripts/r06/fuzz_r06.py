"""Randomised parity run of round 6's new paths on a GPU box, against the CPU oracle:
  * batch tracebacks: 16-byte and packed records, every cost family (unit, unit x g, weighted, affine, transposition), checkpoint route
    with every TILE / STILE incl. 128, sub-batches (TA_TRACE_CHUNK_PAIRS), cut scripts in the packed form;
  * the device set (device 0 listed N times): host-pointer batches (fixed / CSR, k-bounded / exp / hamming) with small chunks and ring slots,
    resident sharded pairs, host searches fanned out over tiny shards (levenshtein + hamming, All / Best), sharded resident haystacks.
usage: python scripts/r06/fuzz_r06.py <minutes> [seed]   (prints the first mismatch and exits 1)"""
import os, sys, time
os.environ.setdefault("TA_TUNING", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import oracle_lib as O
import triple_accel_amd as T
from triple_accel_amd import batch as B
from triple_accel_amd import multi as M
from triple_accel_amd import _native as _N
if os.environ.get("FUZZ_BACKTRACE"): _N.lib().ta_debug_install_abort_backtrace()

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
g = np.random.default_rng(seed)
print("seed", seed, flush=True)
COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (2, 2, 0, None), (3, 3, 0, 3), (2, 3, 1, None), (2, 2, 1, 3), (2, 3, 0, None), (1, 2, 0, None), (3, 2, 1, 2), (1, 1, 1, None)]
ALPHAS = [(33, 127), (97, 99), (0, 1), (0, 13), (12, 14), (0, 256), (65, 69)]
SW = ("TA_TRACE_TILE", "TA_TRACE_STILE", "TA_TRACE_OWN_SWEEP", "TA_TRACE_CSR_OWN_SWEEP", "TA_TRACE_CHUNK_PAIRS", "TA_TRACE_NO_L1", "TA_MULTI_MIN_PAIRS", "TA_MULTI_MIN_HAY",
      "TA_MULTI_CHUNK_BYTES", "TA_MULTI_CHUNK_PAIRS", "TA_MULTI_PIECE", "TA_MULTI_DIRECT_FROM", "TA_MULTI_STAGERS_FROM")


def mutate(x, sym, edits, trans):
    y = bytearray(x)
    for _ in range(edits):
        t = int(g.integers(0, 4 if trans else 3)); pos = int(g.integers(0, len(y) + 1)); c = int(g.integers(sym[0], sym[1]))
        if t == 0 and pos < len(y): y[pos] = c
        elif t == 1: y.insert(pos, c)
        elif t == 2 and pos < len(y): del y[pos]
        elif t == 3 and pos + 1 < len(y): y[pos], y[pos + 1] = y[pos + 1], y[pos]
    return bytes(y)


def pairs(n, lo, hi, sym, edits, trans, sim=0.8):
    a, b = [], []
    for i in range(n):
        x = g.integers(sym[0], sym[1], int(g.integers(lo, hi + 1)), dtype=np.uint8).tobytes()
        y = mutate(x, sym, int(g.integers(0, edits + 2)), trans) if g.random() < sim else g.integers(sym[0], sym[1], int(g.integers(lo, hi + 1)), dtype=np.uint8).tobytes()
        if i & 1: x, y = y, x
        a.append(x); b.append(y)
    return a, b


def fail(what, **kw):
    print("MISMATCH", what, {k: v for k, v in kw.items()}, {s: os.environ.get(s) for s in SW if os.environ.get(s)}, "seed", seed, flush=True)
    sys.exit(1)


SKIP_UNTIL = int(os.environ.get("FUZZ_SKIP_UNTIL", "0"))
SKIP_UNTIL = int(os.environ.get("FUZZ_SKIP_UNTIL", "0"))
t_end, rounds, kinds = time.time() + 60 * minutes, 0, {}
STOP_AT = int(os.environ.get("FUZZ_STOP_AT", "0"))
while time.time() < t_end and not (STOP_AT and rounds >= STOP_AT):
    rounds += 1
    for s in SW: os.environ.pop(s, None)
    kind = int(g.integers(0, 6))
    kinds[kind] = kinds.get(kind, 0) + 1
    costs = COSTS[int(g.integers(0, len(COSTS)))]
    sym = ALPHAS[int(g.integers(0, len(ALPHAS)))]
    trans = costs[3] is not None
    live = rounds >= SKIP_UNTIL            # replaying a seed: earlier rounds only draw their random numbers (in the same order)
    if os.environ.get("FUZZ_VERBOSE") and live: print("round", rounds, "kind", kind, costs, sym, flush=True)
    if kind in (0, 1):                                   # batch tracebacks
        if g.random() < 0.3: os.environ["TA_TRACE_TILE"] = str(g.choice([8, 16, 32]))
        if g.random() < 0.5: os.environ["TA_TRACE_STILE"] = str(g.choice([32, 64, 128]))
        if g.random() < 0.15: os.environ["TA_TRACE_OWN_SWEEP"] = "1"
        if g.random() < 0.15: os.environ["TA_TRACE_CSR_OWN_SWEEP"] = "1"
        if g.random() < 0.3: os.environ["TA_TRACE_CHUNK_PAIRS"] = str(64 * int(g.integers(1, 20)))
        if g.random() < 0.2: os.environ["TA_TRACE_NO_L1"] = "1"
        unit_k = int(g.integers(0, 33))
        k = unit_k * max(costs[0], costs[1]) + costs[2] + int(g.integers(0, 3))
        n = int(g.integers(1, 6000))
        fixed = g.random() < 0.4
        if fixed:
            la, lb = int(g.integers(1, 300)), 0
            lb = max(1, la + int(g.integers(-6, 7)))
            A = g.integers(sym[0], sym[1], (n, la), dtype=np.uint8)
            Bm = np.empty((n, lb), dtype=np.uint8)
            for i in range(n):
                m = (mutate(A[i].tobytes(), sym, int(g.integers(0, unit_k + 2)), trans) + g.integers(sym[0], sym[1], lb, dtype=np.uint8).tobytes())[:lb]
                Bm[i] = np.frombuffer(m, dtype=np.uint8)
            a, b = [r.tobytes() for r in A], [r.tobytes() for r in Bm]
            sa, sb = (B.Strings.from_fixed(A), B.Strings.from_fixed(Bm)) if live else (None, None)
        else:
            a, b = pairs(n, 0, int(g.integers(1, 400)), sym, unit_k, trans)
            sa, sb = (B.Strings.from_list(a), B.Strings.from_list(b)) if live else (None, None)
        packed = kind == 1
        if os.environ.get("FUZZ_VERBOSE") and live: print("  trace n", n, "k", k, "fixed", fixed, {x: os.environ.get(x) for x in SW if os.environ.get(x)}, flush=True)
        cap = None if g.random() < 0.7 else int(g.integers(1, 12))
        if not live:
            g.integers(0, max(1, n // 250))               # (the sample offset drawn below)
            continue
        if packed:
            o, p, ne = B.levenshtein_trace_batch_packed(sa, sb, k, costs, cap=cap)
            scripts, nn = B.packed_to_lists(p, ne, allow_cut=True), ne.cpu().numpy()
        else:
            o, e, ne = B.levenshtein_trace_batch(sa, sb, k, costs, cap=cap)
            scripts, nn = B.edits_to_lists(e, ne, allow_cut=True), ne.cpu().numpy()
        d = o.cpu().numpy().view(np.uint32)
        step = max(1, n // 250)
        for i in range(int(g.integers(0, step)), n, step):
            wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, costs)
            if wd is None:
                if d[i] != 0xFFFFFFFF or scripts[i] != [] or nn[i] != 0: fail("trace none", i=i, a=a[i], b=b[i], k=k, costs=costs, packed=packed)
                continue
            capv = (p.shape[1] if packed else e.shape[1])
            want = we if len(we) <= capv else (we[len(we) - capv:] if packed else we[:capv])
            if d[i] != wd or nn[i] != len(we) or scripts[i] != want:
                fail("trace", i=i, a=a[i], b=b[i], k=k, costs=costs, packed=packed, got=(int(d[i]), scripts[i]), want=(wd, want), fixed=fixed)
    elif kind in (2, 3):                                 # the device set: pair batches
        world = int(g.choice([1, 2, 3, 5, 8]))
        if live: M.set_devices([0] * world)
        os.environ["TA_MULTI_MIN_PAIRS"] = str(int(g.choice([1, 16, 100, 4096])))
        os.environ["TA_MULTI_CHUNK_BYTES"] = str(int(g.choice([2048, 30000, 1 << 20, 64 << 20])))
        os.environ["TA_MULTI_PIECE"] = str(int(g.choice([4096, 65536, 4 << 20])))
        os.environ["TA_MULTI_DIRECT_FROM"] = str(int(g.choice([1, 4096, 65536, 1 << 40])))
        os.environ["TA_MULTI_STAGERS_FROM"] = str(int(g.choice([1, 50000, 16 << 20])))
        if g.random() < 0.3: os.environ["TA_MULTI_CHUNK_PAIRS"] = str(int(g.integers(1, 500)))
        n = int(g.integers(0, 4000))
        unit_k = int(g.integers(0, 40))
        if os.environ.get("FUZZ_VERBOSE") and (live or rounds >= STOP_AT - 3): print("  pairs world", world, "n", n, {x: os.environ.get(x) for x in SW if os.environ.get(x)}, flush=True)
        k = unit_k * max(costs[0], costs[1]) + costs[2]
        if g.random() < 0.4 and n:
            la = int(g.integers(1, 200))
            A = g.integers(sym[0], sym[1], (n, la), dtype=np.uint8); Bm = A.copy()
            pos = g.integers(0, la, size=(n, max(1, unit_k // 2))); Bm[np.arange(n)[:, None], pos] = sym[0]
            ha, hb, ca, cb = A, Bm, O.csr_from_fixed(A), O.csr_from_fixed(Bm)
        else:
            a, b = pairs(n, 0, int(g.integers(1, 300)), sym, unit_k, trans)
            ha, hb, ca, cb = a, b, O.csr_from_list(a), O.csr_from_list(b)
        if not live:
            if kind == 2:
                if n: g.random()
            else:
                g.random()
            continue
        if kind == 2:
            got, want = M.levenshtein_k_batch_host(ha, hb, k, costs), O.levenshtein_k_batch(ca, cb, k, costs)
            if not np.array_equal(got, want): fail("k_batch_host", world=world, n=n, k=k, costs=costs)
            if n and g.random() < 0.5:
                S = M.ShardedPairs(ha, hb)
                if not np.array_equal(S.levenshtein_k(k, costs), want): fail("sharded pairs", world=world, n=n)
                S.close()
        else:
            if g.random() < 0.5:
                got, want = M.levenshtein_exp_batch_host(ha, hb, costs), O.levenshtein_exp_batch(ca, cb, costs)
            else:
                got, want = M.hamming_batch_host(ha, hb), O.hamming_batch(ca, cb)
            if not np.array_equal(got, want): fail("exp / hamming batch host", world=world, n=n, costs=costs)
        M.set_devices([0])
    else:                                                # the device set: searches
        world = int(g.choice([2, 3, 5, 8]))
        if live: M.set_devices([0] * world)
        h = int(g.integers(100, 60000))
        os.environ["TA_MULTI_MIN_HAY"] = str(int(g.choice([4, 50, 1000, 8000])))
        os.environ["TA_MULTI_PIECE"] = str(int(g.choice([4096, 65536])))
        s2 = sym if sym[1] - sym[0] > 3 else (33, 127)
        nl = int(g.integers(1, 40))
        needle = g.integers(max(1, s2[0]), s2[1], nl, dtype=np.uint8).tobytes()
        hay = bytearray(g.integers(max(1, s2[0]), s2[1], h, dtype=np.uint8).tobytes())
        for pos in range(int(g.integers(0, 500)), h - 2 * nl, int(g.integers(300, 5000))):
            m = mutate(needle, (max(1, s2[0]), s2[1]), int(g.integers(0, 4)), trans)
            hay[pos:pos + len(m)] = m
        hay = bytes(hay)
        if os.environ.get("FUZZ_VERBOSE") and live: print("  search world", world, "h", h, "nl", nl, {x: os.environ.get(x) for x in SW if os.environ.get(x)}, flush=True)
        if kind == 4:
            sc = costs if O.costs_valid_search(costs) else (1, 1, 0, None)
            k = int(g.integers(0, max(1, nl // 2) + 1)) * max(sc[0], sc[1])
            if not live:
                if g.random() < 0.5: g.integers(nl + k + 2, nl + k + 300)
                continue
            for st in (T.SearchType.All, T.SearchType.Best):
                got = [tuple(m) for m in T.levenshtein_search_simd_with_opts(needle, hay, k, st, T.EditCosts(*sc), False)]
                want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, sc, False)
                if got != want: fail("search host", world=world, needle=needle, h=h, k=k, costs=sc, st=st, got=got[:5], want=want[:5])
            if g.random() < 0.5:
                H = M.ShardedHaystack(hay, overlap=int(g.integers(nl + k + 2, nl + k + 300)))
                got = [tuple(m) for m in H.levenshtein_search(needle, k, T.SearchType.Best, sc)]
                if got != O.levenshtein_search_naive_with_opts(needle, hay, k, O.BEST, sc, False): fail("sharded haystack", world=world, needle=needle, k=k)
                H.close()
        else:
            k = int(g.integers(0, nl // 2 + 1))
            if g.random() < 0.2:
                hz = bytearray(hay); hz[int(g.integers(0, h))] = 0; hay = bytes(hz)
            if not live: continue
            for st in (T.SearchType.All, T.SearchType.Best):
                try:
                    want = O.hamming_search_simd_with_opts(needle, hay, k, st)
                except ValueError:
                    want = "panic"
                try:
                    got = [tuple(m) for m in T.hamming_search_simd_with_opts(needle, hay, k, st)]
                except T.PanicError:
                    got = "panic"
                if got != want: fail("hamming search host", world=world, needle=needle, h=h, k=k, st=st)
                gotn = [tuple(m) for m in T.hamming_search_naive_with_opts(needle, hay, k, st)]
                if gotn != O.hamming_search_naive_with_opts(needle, hay, k, st): fail("hamming naive search host", world=world, needle=needle, h=h, k=k, st=st)
        M.set_devices([0])
print("ok: %d rounds, kinds %s, seed %d" % (rounds, sorted(kinds.items()), seed), flush=True)
