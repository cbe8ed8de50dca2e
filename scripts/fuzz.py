"""Randomised parity run on a GPU box: random batches through every distance / search entry point against the CPU oracle.
usage: python scripts/fuzz.py <minutes> [seed]   (not part of the test suite; prints the first mismatch and exits 1)"""
import os, sys, time
os.environ.setdefault("TA_TUNING", "1")      # lets a round switch the VLINE fetch form on (TA_BITS_VLINE is read per call under TA_TUNING)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen as Dg
import oracle_lib as O
import triple_accel_amd as T
from triple_accel_amd import batch as B

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
g = np.random.default_rng(seed)
print("seed", seed, flush=True)
COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (2, 1, 0, None), (1, 2, 0, None), (3, 2, 1, 2), (2, 3, 2, None), (1, 1, 1, None), (5, 3, 0, 4), (2, 2, 1, 3), (255, 255, 0, None), (2, 2, 0, None), (3, 3, 0, 3)]


def rand_pairs(n, lo, hi, alpha, sim, edits):
    a, b = [], []
    for _ in range(n):
        la = int(g.integers(lo, hi + 1))
        x = g.integers(alpha[0], alpha[1], la, dtype=np.uint8).tobytes()
        if g.random() < sim:
            y = Dg.mutate(Dg.rng(int(g.integers(1 << 30))), x, int(g.integers(0, edits + 1)), True) if la else x
        else:
            y = g.integers(alpha[0], alpha[1], int(g.integers(lo, hi + 1)), dtype=np.uint8).tobytes()
        a.append(x); b.append(y)
    return a, b


t_end, rounds, kinds = time.time() + 60 * minutes, 0, {}
while time.time() < t_end:
    rounds += 1
    kind = int(g.integers(0, 13))
    os.environ.pop("TA_BITS_VLINE", None)
    if g.random() < 0.3:
        os.environ["TA_BITS_VLINE"] = "1"                # round 4: CSR batches through the VLINE fetch form
    early = bool(g.random() < 0.3)                 # the early-out option must never change an answer
    T.set_option(T.OPT_EARLY_OUT, early)
    T.set_option(T.OPT_UNIT_PREFILTER, bool(g.random() < 0.4))      # round 5: nor must the unit-cost pre-pass of weighted batches
    for sw in ("TA_EXP_HOST_ROUNDS", "TA_TRACE_TILE", "TA_TRACE_STILE", "TA_HAMMING_PHASE_Q", "TA_TRACE_NO_BITS"):
        os.environ.pop(sw, None)
    if g.random() < 0.2:
        os.environ["TA_EXP_HOST_ROUNDS"] = "1"           # the host-driven levenshtein_exp rounds (big batches are device-driven by default)
    if g.random() < 0.5:
        os.environ["TA_TRACE_TILE"] = str(g.choice([8, 16, 32])); os.environ["TA_TRACE_STILE"] = str(g.choice([32, 64]))
    if g.random() < 0.3:
        os.environ["TA_HAMMING_PHASE_Q"] = str(g.choice([1, 2]))
    alpha = [(1, 256), (97, 101), (0, 256), (12, 14), (33, 127)][int(g.integers(0, 5))]
    costs = COSTS[int(g.integers(0, len(COSTS)))]
    if not O.costs_valid(costs):
        continue
    try:
        if kind in (0, 1):      # k-bounded batch, ragged or fixed, many geometries
            n = int(g.choice([1, 7, 64, 500, 1500, 4000]))
            hi = int(g.choice([8, 40, 130, 300, 700, 2500]))
            lo = hi if kind == 1 else int(g.integers(0, hi + 1))
            a, b = rand_pairs(n, lo, hi, alpha, 0.7, int(g.choice([2, 10, 40, 150])))
            k = int(g.choice([0, 1, 3, 8, 20, 32, 33, 47, 64, 100, 127, 128, 300, 2000, 0xFFFFFFFF]))
            if kind == 1 and all(len(x) == len(a[0]) for x in a) and all(len(y) == len(b[0]) for y in b) and len(a[0]) and len(b[0]):
                fa = np.frombuffer(b"".join(a), dtype=np.uint8).reshape(n, -1); fb = np.frombuffer(b"".join(b), dtype=np.uint8).reshape(n, -1)
                got = B.levenshtein_k_batch(B.Strings.from_fixed(fa), B.Strings.from_fixed(fb), k, costs).cpu().numpy().view(np.uint32)
            else:
                got = B.levenshtein_k_batch(B.Strings.from_list(a), B.Strings.from_list(b), k, costs).cpu().numpy().view(np.uint32)
            want = O.levenshtein_k_batch(O.csr_from_list(a), O.csr_from_list(b), k, costs)
            ok = np.array_equal(got, want)
            what = ("k_batch", n, lo, hi, k, costs, alpha, T.last_launch_info()["kernel"])
        elif kind in (2, 11):   # exp batch (11, round 5: batches of >= 1024 pairs -- the device-driven rounds)
            n = int(g.choice([1, 30, 64, 300])) if kind == 2 else int(g.choice([1024, 1500, 5000]))
            hi = int(g.choice([20, 100, 400, 1200]))
            a, b = rand_pairs(n, int(g.integers(0, hi + 1)), hi, alpha, 0.6, int(g.choice([3, 30, 200])))
            got = B.levenshtein_exp_batch(B.Strings.from_list(a), B.Strings.from_list(b), costs).cpu().numpy().view(np.uint32)
            want = O.levenshtein_exp_batch(O.csr_from_list(a), O.csr_from_list(b), costs)
            ok = np.array_equal(got, want)
            what = ("exp_batch", n, hi, costs, alpha)
        elif kind == 3:         # single calls incl. traceback
            a, b = rand_pairs(1, 0, int(g.choice([10, 80, 300, 1500])), alpha, 0.8, 30)
            x, y = a[0], b[0]
            k = int(g.choice([0, 2, 10, 40, 200, 0xFFFFFFFF]))
            tr = bool(g.integers(0, 2))
            got = T.levenshtein_simd_k_with_opts(x, y, k, tr, T.EditCosts(*costs))
            want = O.levenshtein_simd_k_with_opts(x, y, k, tr, costs)
            ok = (got is None and want[0] is None) or (got is not None and want[0] is not None and got[0] == want[0] and (not tr or [tuple(e) for e in got[1]] == [tuple(e) for e in want[1]]))
            what = ("single", len(x), len(y), k, tr, costs)
        elif kind == 4:         # levenshtein search
            if not O.costs_valid_search(costs):
                continue
            n = int(g.choice([1, 5, 17, 32, 60, 200]))
            needle = g.integers(max(1, alpha[0]), alpha[1], n, dtype=np.uint8).tobytes()
            k = int(g.integers(0, max(1, n // 2) + 2))
            hay = Dg.planted_haystack(int(g.integers(1 << 30)), needle, int(g.choice([500, 20000, 300000])), int(g.choice([50, 2000])), max(1, k))
            st = int(g.integers(0, 2)); anch = bool(g.random() < 0.2)
            got = [tuple(m) for m in T.levenshtein_search_simd_with_opts(needle, hay, k, st, T.EditCosts(*costs), anch)]
            want = O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, anch)
            ok = got == want
            what = ("search", n, k, st, anch, costs, len(hay))
        elif kind in (6, 7):    # round 3: big batches -- two pairs per lane, length-ordered CSR, the small-alphabet kernel (with foreign bytes)
            unit = [(1, 1, 0, None), (1, 1, 0, 1)][int(g.integers(0, 2))]
            costs = unit if kind == 7 or g.random() < 0.7 else costs
            small = kind == 7
            n = int(g.choice([20000, 70000, 270000]))
            L = int(g.choice([24, 64, 128, 200, 256, 300]))
            Lb = max(1, L + int(g.integers(-6, 7)))
            k = int(g.choice([0, 3, 8, 12, 14, 15, 20, 30, 32, 33]))
            small_alphabet = [b"ACGT", b"ACGT", b"ACGTN", b"ACGTRYSWKMBDHVNU", b"ACDEFGHIKLMNPQRSTVWY", b"0123456789", b"acgu"][int(g.integers(0, 7))]
            sym = np.frombuffer(small_alphabet, dtype=np.uint8) if small else np.arange(alpha[0], alpha[1], dtype=np.uint8)
            fa = sym[g.integers(0, len(sym), size=(n, L))]
            fb = sym[g.integers(0, len(sym), size=(n, Lb))]
            near = g.random(n) < 0.5
            m = min(L, Lb)
            fb[near, :m] = fa[near, :m]
            pos = g.integers(0, m, size=(n, 4))
            rows = np.nonzero(near)[0]
            fb[rows[:, None], pos[rows]] = sym[g.integers(0, len(sym), size=(len(rows), 4))]
            alphabet = None
            if small:
                alphabet = small_alphabet
                os.environ["TA_BITSQ_WIDE"] = "1" if g.random() < 0.6 else "0"      # the 5-bit-code kernel wherever it can run / where it pays
                bad = g.choice(n, size=int(g.integers(0, 50)), replace=False)
                outside = np.array([x for x in range(256) if x not in small_alphabet], dtype=np.uint8)
                (fa if g.random() < 0.5 else fb)[bad, g.integers(0, m, size=len(bad))] = outside[g.integers(0, len(outside), size=len(bad))]
            if g.random() < 0.5 and not small:              # the same pairs as a CSR batch with ragged tails: taken in length order
                la = g.integers(max(1, L - 40), L + 1, size=n); lb = np.minimum(Lb, np.maximum(1, la + g.integers(-5, 6, size=n)))
                a = [fa[i, :la[i]].tobytes() for i in range(n)]; b = [fb[i, :lb[i]].tobytes() for i in range(n)]
                got = B.levenshtein_k_batch(B.Strings.from_list(a), B.Strings.from_list(b), k, costs).cpu().numpy().view(np.uint32)
                want = O.levenshtein_k_batch(O.csr_from_list(a), O.csr_from_list(b), k, costs)
            else:
                got = B.levenshtein_k_batch(B.Strings.from_fixed(fa), B.Strings.from_fixed(fb), k, costs, alphabet=alphabet).cpu().numpy().view(np.uint32)
                want = O.levenshtein_k_batch(O.csr_from_fixed(fa), O.csr_from_fixed(fb), k, costs)
            ok = np.array_equal(got, want)
            what = ("big_batch", n, L, Lb, k, costs, small, early, T.last_launch_info()["kernel"], T.last_kernel_name())
        elif kind in (9, 12):   # round 4: batch tracebacks on the device; 12, round 5: the unit-cost families on the checkpoint-and-recompute kernel
            n = int(g.choice([1, 40, 700, 3000]))
            hi = int(g.choice([12, 60, 200, 500]))
            if kind == 12:
                costs = [(1, 1, 0, None), (1, 1, 0, 1)][int(g.integers(0, 2))]
                hi = int(g.choice([12, 60, 200, 500, 1400]))
            a, b = rand_pairs(n, int(g.integers(0, hi + 1)), hi, alpha, 0.8, int(g.choice([2, 10, 30])))
            k = int(g.choice([0, 3, 10, 32, 64, 200])) if kind == 9 else int(g.choice([0, 1, 3, 10, 20, 30, 32]))
            out, ed, ne = B.levenshtein_trace_batch(B.Strings.from_list(a), B.Strings.from_list(b), k, costs)
            gd = out.cpu().numpy().view(np.uint32); ge = B.edits_to_lists(ed, ne)
            ok = True
            for i in range(n):
                wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, costs)
                if wd is None:
                    ok = ok and gd[i] == 0xFFFFFFFF and ge[i] == []
                else:
                    ok = ok and gd[i] == wd and ge[i] == [tuple(e) for e in we]
                if not ok:
                    print("pair", i, a[i], b[i], gd[i], ge[i], wd, we)
                    break
            what = ("trace_batch", n, hi, k, costs, alpha, T.last_kernel_name())
        elif kind == 10:        # round 4: hamming_search kernels by (needle length, k): SWAR16, bit-sliced, the long-needle form
            n = int(g.choice([1, 3, 4, 8, 9, 12, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64, 65, 100, 300]))
            lo_c = 97 if g.random() < 0.3 else 1                # (a four-letter text now and then: the subset filters pass many candidates)
            hi_c = 101 if lo_c == 97 else 256
            needle = g.integers(lo_c, hi_c, n, dtype=np.uint8).tobytes()
            hay = bytearray(g.integers(lo_c, hi_c, int(g.choice([n, 700, 40000, 600000])), dtype=np.uint8).tobytes())
            for pos in range(0, max(1, len(hay) - n), max(n + 3, 3000)):
                m = bytearray(needle)
                for _ in range(int(g.integers(0, 9))):
                    m[int(g.integers(0, n))] = int(g.integers(lo_c, hi_c))
                hay[pos:pos + n] = m
            hay = bytes(hay[: max(len(hay), n)])
            k = int(g.choice([0, 1, 2, 3, 4, 7, 8, 15, 16, 31, 32, n]))
            got = [tuple(int(v) for v in r) for r in B.hamming_search_dev(needle, B.haystack_tensor(hay), k)]
            want = O.hamming_search_naive_with_opts(needle, hay, k, O.ALL)
            ok = got == want
            what = ("hamming_kernels", n, k, len(hay), T.last_kernel_name())
        elif kind == 8:         # first hit of a long haystack (the lazy All-mode iterator's first element)
            if not O.costs_valid_search(costs):
                continue
            n = int(g.choice([4, 17, 32, 60]))
            needle = g.integers(max(1, alpha[0]), alpha[1], n, dtype=np.uint8).tobytes()
            k = int(g.integers(0, max(1, n // 3) + 1))
            size = int(g.choice([3000, 70000, 300000, 1500000]))
            hay = bytearray(g.integers(max(1, alpha[0]), alpha[1], size, dtype=np.uint8).tobytes())
            if g.random() < 0.8:
                p0 = int(g.integers(0, size - n))
                hay[p0:p0 + n] = needle
            hay = bytes(hay)
            got = T.levenshtein_search_first(needle, hay, k, T.EditCosts(*costs))
            want = O.levenshtein_search_naive_with_opts(needle, hay, k, O.ALL, costs, False)
            ok = (tuple(got) if got else None) == (want[0] if want else None)
            what = ("search_first", n, k, costs, size)
        else:                   # hamming + hamming search
            n = int(g.choice([1, 9, 32, 33, 100, 700]))
            needle = g.integers(1, 256, n, dtype=np.uint8).tobytes()
            hay = bytearray(g.integers(1, 256, int(g.choice([n, 1000, 200000])), dtype=np.uint8).tobytes())
            for pos in range(0, max(1, len(hay) - n), max(n + 3, 5000)):
                m = bytearray(needle)
                for _ in range(int(g.integers(0, 4))):
                    m[int(g.integers(0, n))] = int(g.integers(1, 256))
                hay[pos:pos + n] = m
            hay = bytes(hay[: max(len(hay), n)])
            k = int(g.integers(0, n // 2 + 2)); st = int(g.integers(0, 2))
            got = [tuple(m) for m in T.hamming_search_simd_with_opts(needle, hay, k, st)]
            want = O.hamming_search_naive_with_opts(needle, hay, k, st)
            ok = got == want and T.hamming(needle, hay[:n]) == O.hamming_naive(needle, hay[:n])
            what = ("hamming_search", n, k, st, len(hay))
    except Exception as e:
        print("EXCEPTION", kind, costs, alpha, repr(e)[:300], "last launch", T.last_launch_info(), flush=True)
        import traceback; traceback.print_exc()
        try:
            print("params:", n, hi, lo if kind in (0, 1) else None, k if kind != 2 else None, flush=True)
        except Exception:
            pass
        sys.exit(1)
    kinds[what[0]] = kinds.get(what[0], 0) + 1
    if not ok:
        print("MISMATCH", what, flush=True)
        sys.exit(1)
print("fuzz ok:", rounds, "rounds", kinds, flush=True)
