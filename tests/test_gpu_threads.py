"""-m gpu: the C ABI under concurrent callers.  Eight Python threads (ctypes drops the GIL across a foreign call, so the
library really runs concurrently) hammer the single-call entry points -- ta_levenshtein_simd_k_with_opts with mixed cost
sets and lengths, ta_levenshtein_search, ta_hamming, a traceback now and then -- and every answer must equal the
oracle's.  Reference contract: pure re-entrant functions (src/levenshtein.rs:714-720, :2508-2511; SURVEY.md 8b Threading)."""
import threading

import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O
import triple_accel_amd as T

pytestmark = pytest.mark.gpu

COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (2, 1, 1, None), (3, 2, 0, 2)]


def _work(tid, rounds, errors):
    try:
        g = Dg.rng(1000 + tid)
        for r in range(rounds):
            n = int(g.integers(0, 400 if r % 7 else 5000))
            a = Dg.rand_str(g, n)
            b = Dg.mutate(g, a, int(g.integers(0, 24)), swaps=True) if g.random() < 0.7 else Dg.rand_str(g, int(g.integers(0, 400)))
            costs = COSTS[int(g.integers(0, len(COSTS)))]
            k = int(g.integers(0, 60))
            want = O.levenshtein_naive_k_with_opts(a, b, k, False, costs)
            got = T.levenshtein_simd_k_with_opts(a, b, k, False, T.EditCosts(*costs))
            assert (None if got is None else got[0]) == want[0], ("k", tid, r, n, k, costs)
            if r % 5 == 0 and len(a) == len(b):
                assert T.hamming(a, b) == O.hamming_naive(a, b)
            if r % 4 == 0:
                needle = Dg.rand_str(g, int(g.integers(1, 20)))
                hay = Dg.planted_haystack(int(g.integers(1 << 30)), needle, int(g.integers(50, 3000)), 120, 3)
                ks = int(g.integers(0, 5))
                st = O.BEST if r % 8 else O.ALL
                w = O.levenshtein_search_naive_with_opts(needle, hay, ks, st, COSTS[0], False)
                gg = T.levenshtein_search_simd_with_opts(needle, hay, ks, T.SearchType.Best if st == O.BEST else T.SearchType.All,
                                                          T.LEVENSHTEIN_COSTS, False)
                assert [tuple(m) for m in gg] == w, ("search", tid, r)
            if r % 9 == 0 and n < 300:
                wt = O.levenshtein_naive_k_with_opts(a, b, 64, True, costs)
                gt = T.levenshtein_simd_k_with_opts(a, b, 64, True, T.EditCosts(*costs))
                assert (wt[0] is None) == (gt is None), ("trace", tid, r)
                if gt is not None:
                    assert gt[0] == wt[0] and [(e.edit, e.count) for e in gt[1]] == [(nm, c) for nm, c in wt[1]], ("trace", tid, r)
    except BaseException as e:      # noqa: BLE001 -- reported by the main thread
        errors.append((tid, repr(e)))


def test_eight_threads_against_the_oracle():
    errors = []
    ts = [threading.Thread(target=_work, args=(t, 120, errors)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in ts), "a caller thread hung"
    assert not errors, errors[:3]


def test_stream_switch_keeps_scratch_ordered():
    """One thread, two torch streams, batch calls that keep kernel-side state in thread-local scratch (the exp loop's subset
    lists): the second call must wait for the first (event recorded at the end of every call), results stay the oracle's."""
    import torch
    from triple_accel_amd import batch as B
    a, b = Dg.pairs_mutated_fixed(77, 3000, 200, 40)
    sa, sb = B.Strings.from_fixed(a), B.Strings.from_fixed(b)
    want = O.levenshtein_exp_batch(O.csr_from_fixed(a), O.csr_from_fixed(b), (1, 1, 0, None))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(6):
        with torch.cuda.stream(s1 if rep % 2 == 0 else s2):
            outs.append(B.levenshtein_exp_batch(sa, sb, (1, 1, 0, None)))
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy().view(np.uint32), want)


def test_thread_release_is_idempotent_and_reusable():
    from triple_accel_amd import _native
    assert T.levenshtein(b"kitten", b"sitting") == 3
    _native.lib().ta_thread_release()
    _native.lib().ta_thread_release()
    assert T.levenshtein(b"kitten", b"sitting") == 3 and T.hamming(b"abc", b"abd") == 1


def _batch_work(tid, rounds, errors):
    """One thread of test_four_threads_on_the_batch_entries: its own stream, its own options, this round's batch paths."""
    import torch
    from triple_accel_amd import batch as B
    try:
        g = Dg.rng(7000 + tid)
        st = torch.cuda.Stream()
        T.set_option(T.OPT_UNIT_PREFILTER, tid % 2 == 1)                 # options are per thread
        with torch.cuda.stream(st):
            for r in range(rounds):
                n = int(g.choice([64, 1100, 2500]))
                L = int(g.choice([40, 120, 200]))
                am, bm = Dg.pairs_mutated_fixed(int(g.integers(1 << 30)), n, L, int(g.integers(2, 12)), swaps=True)
                bm[::3] = Dg.pairs_random(int(g.integers(1 << 30)), len(bm[::3]), L)[1]
                sa, sb = B.Strings.from_fixed(am), B.Strings.from_fixed(bm)
                ca, cb = O.csr_from_fixed(am), O.csr_from_fixed(bm)
                what = r % 4
                if what == 0:                                            # levenshtein_exp over a batch (device-driven rounds from 1,024 pairs)
                    costs = [(1, 1, 0, None), (1, 1, 0, 1), (2, 3, 1, None)][int(g.integers(0, 3))]
                    got = B.levenshtein_exp_batch(sa, sb, costs).cpu().numpy().view(np.uint32)
                    assert np.array_equal(got, O.levenshtein_exp_batch(ca, cb, costs)), ("exp", tid, r, n, L, costs)
                elif what == 1:                                          # tracebacks of a batch: checkpoint kernel (fixed-length: folded sweep)
                    costs = [(1, 1, 0, None), (1, 1, 0, 1)][int(g.integers(0, 2))]
                    k = int(g.choice([6, 20, 30]))
                    out, ed, ne = B.levenshtein_trace_batch(sa, sb, k, costs)
                    d, scripts = out.cpu().numpy().view(np.uint32), B.edits_to_lists(ed, ne)
                    for i in range(0, n, max(1, n // 40)):
                        wd, we = O.levenshtein_simd_k_with_opts(am[i].tobytes(), bm[i].tobytes(), k, True, costs)
                        assert (d[i] == wd and scripts[i] == we) if wd is not None else (d[i] == 0xFFFFFFFF and scripts[i] == []), ("trace", tid, r, i)
                elif what == 2:                                          # weighted k-bounded batch (odd threads: behind the unit-cost pre-pass)
                    costs, k = [((2, 3, 1, None), 30), ((2, 2, 1, 3), 12), ((3, 1, 0, None), 20)][int(g.integers(0, 3))]
                    got = B.levenshtein_k_batch(sa, sb, k, costs).cpu().numpy().view(np.uint32)
                    assert np.array_equal(got, O.levenshtein_k_batch(ca, cb, k, costs)), ("k_batch", tid, r, costs)
                else:                                                    # hamming_search on a resident haystack (report through pinned memory)
                    nl = int(g.choice([8, 24, 32, 64]))
                    needle = bytes(int(c) or 1 for c in Dg.random_bytes(g, nl))
                    hay = Dg.random_bytes(g, 200_000)
                    hay[hay == 0] = 1
                    for pos in range(1000, hay.size - 2 * nl, 30_011):
                        hay[pos:pos + nl] = np.frombuffer(needle, dtype=np.uint8)
                        hay[pos + int(g.integers(0, nl))] = 9
                    k = nl // 4
                    got = [tuple(int(v) for v in row) for row in B.hamming_search_dev(needle, B.haystack_tensor(hay), k)]
                    assert got == O.hamming_search_naive_with_opts(needle, hay.tobytes(), k, O.ALL), ("hsearch", tid, r, nl)
        torch.cuda.synchronize()
    except BaseException as e:      # noqa: BLE001 -- reported by the main thread
        errors.append((tid, repr(e)[:400]))
    finally:
        T.set_option(T.OPT_UNIT_PREFILTER, False)


def test_four_threads_on_the_batch_entries():
    """Round 5's batch paths under concurrent callers, each on its own stream with its own options: device-driven levenshtein_exp rounds,
    checkpoint tracebacks, weighted batches with and without the unit-cost pre-pass, hamming_search with its pinned report box -- all of
    them keep per-thread scratch, and every answer must be the oracle's."""
    errors = []
    ts = [threading.Thread(target=_batch_work, args=(t, 16, errors)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=900)
    assert not any(t.is_alive() for t in ts), "a caller thread hung"
    assert not errors, errors[:3]
