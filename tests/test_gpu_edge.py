"""-m gpu: edge cases -- empty and tiny inputs, maximum costs, very long strings with a narrow band, big batches."""
import os

import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O

pytestmark = pytest.mark.gpu


def gpu_k(a_list, b_list, k, costs=(1, 1, 0, None)):
    from triple_accel_amd import batch as B
    out = B.levenshtein_k_batch(B.Strings.from_list(a_list), B.Strings.from_list(b_list), k, costs)
    return out.cpu().numpy().view(np.uint32)


def oracle_k(a_list, b_list, k, costs=(1, 1, 0, None)):
    return O.levenshtein_k_batch(O.csr_from_list(a_list), O.csr_from_list(b_list), k, costs)


def test_empty_batch_and_single_pairs():
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    assert gpu_k([], [], 3).size == 0
    assert T.levenshtein(b"", b"") == 0 and T.levenshtein(b"", b"abc") == 3 and T.levenshtein(b"a", b"") == 1
    assert T.levenshtein_simd_k(b"", b"abc", 2) is None
    assert T.hamming(b"", b"") == 0
    for n in (1, 15, 16, 17, 63, 64, 65):
        x = bytes(range(1, n + 1)); y = bytes((v + (i % 3 == 0)) & 0xFF or 1 for i, v in enumerate(x))
        assert T.hamming(x, y) == sum(p != q for p, q in zip(x, y))
    assert list(T.levenshtein_search(b"abc", b"")) == []
    assert list(T.levenshtein_search(b"abcdef", b"ab")) == [tuple(m) for m in O.levenshtein_search_naive_with_opts(b"abcdef", b"ab", 3, O.BEST)]
    assert [tuple(m) for m in T.levenshtein_search(b"a", b"xxaxx")] == O.levenshtein_search_naive_with_opts(b"a", b"xxaxx", 1, O.BEST)


def test_k_zero_and_max_costs():
    g = Dg.rng(1)
    a = [Dg.rand_str(g, int(g.integers(0, 30))) for _ in range(500)]
    b = [x if i % 2 else Dg.mutate(g, x, 2) for i, x in enumerate(a)]
    assert np.array_equal(gpu_k(a, b, 0), oracle_k(a, b, 0))
    for costs in [(255, 255, 255, None), (255, 255, 0, 255), (1, 255, 255, None), (255, 1, 0, 1)]:
        assert O.costs_valid(costs)
        for k in (0, 255, 1000, 70000, 0xFFFFFFFF):
            assert np.array_equal(gpu_k(a, b, k, costs), oracle_k(a, b, k, costs)), (costs, k)


def test_very_long_strings_narrow_band():
    """200 KB strings, k = 24: thousands of stream chunks per pair, band of 51 diagonals."""
    g = Dg.rng(2)
    a, b = [], []
    for _ in range(6):
        x = Dg.rand_str(g, 200_000)
        a.append(x); b.append(Dg.mutate(g, x, 20))
    a.append(a[0]); b.append(Dg.rand_str(g, 200_010))
    got, want = gpu_k(a, b, 24), oracle_k(a, b, 24)
    assert np.array_equal(got, want) and (want[:6] != 0xFFFFFFFF).all() and want[6] == 0xFFFFFFFF


def test_big_batch_tiny_strings():
    """3M pairs of 0..12-byte strings (many waves, ragged, lots of empties)."""
    g = Dg.rng(3)
    n = 3_000_000
    la = g.integers(0, 13, size=n); lb = g.integers(0, 13, size=n)
    blob_a = g.integers(97, 101, size=int(la.sum()) + 16, dtype=np.uint8)
    blob_b = g.integers(97, 101, size=int(lb.sum()) + 16, dtype=np.uint8)
    off_a = np.concatenate([[0], np.cumsum(la)]).astype(np.uint64); off_b = np.concatenate([[0], np.cumsum(lb)]).astype(np.uint64)
    import torch
    from triple_accel_amd import batch as B
    sa = B.Strings(torch.from_numpy(blob_a).cuda(), torch.from_numpy(off_a.astype(np.int64)).cuda(), max_len=12)
    sb = B.Strings(torch.from_numpy(blob_b).cuda(), torch.from_numpy(off_b.astype(np.int64)).cuda(), max_len=12)
    got = B.levenshtein_k_batch(sa, sb, 5, (1, 1, 0, 1)).cpu().numpy().view(np.uint32)
    want = O.levenshtein_k_batch((blob_a, off_a), (blob_b, off_b), 5, (1, 1, 0, 1))
    assert np.array_equal(got, want)


def test_first_call_before_any_torch_use_in_a_fresh_process():
    """A fresh interpreter whose FIRST GPU action is a single call of this package (torch untouched until then) must work,
    and torch must still see the GPU afterwards (the loader brings torch's HIP runtime up first: _native.lib())."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import triple_accel_amd as T; "
            "assert T.levenshtein(b'kitten', b'sitting') == 3; assert T.device_count() >= 1; "
            "from triple_accel_amd import batch as B; import numpy as np, torch; "
            "a = np.full((70, 40), 65, dtype=np.uint8); "
            "out = B.levenshtein_k_batch(B.Strings.from_fixed(a), B.Strings.from_fixed(a), 3).cpu().numpy(); "
            "assert (out == 0).all() and torch.cuda.is_available(); print('fresh ok')" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fresh ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_generic_items_beyond_256_symbols_and_buffer_inputs():
    """levenshtein_naive<T: PartialEq> (src/levenshtein.rs:105-148) on sequences with far more than 256 distinct items: only the
    items common to both sides need codes of their own.  Inputs with the buffer protocol (array.array, ctypes arrays) are bytes."""
    import array
    import ctypes
    import triple_accel_amd as T
    g = Dg.rng(77)
    for trial in range(8):
        n = int(g.integers(300, 600))
        # ~n distinct items in a; b shares the items of a 150-symbol pool with it and replaces the rest with items of its own:
        # far more than 256 distinct items in the pair, at most 150 of them common to both sides
        pool = [int(v) for v in g.integers(0, 150, size=n)]
        a = [pool[i] if i % 3 else (1 << 20) + i for i in range(n)]
        b = [pool[i] if i % 3 else (1 << 21) + i for i in range(n)]
        for _ in range(int(g.integers(0, 12))):
            p = int(g.integers(0, len(b)))
            r = int(g.integers(0, 3))
            if r == 0: b[p] = (1 << 22) + p
            elif r == 1: b.insert(p, pool[p % n])
            else: del b[p]
        assert len(set(a) | set(b)) > 300
        # reference value: a plain two-row DP over the items themselves
        prev = list(range(len(a) + 1))
        for j in range(1, len(b) + 1):
            cur = [j] + [0] * len(a)
            for i in range(1, len(a) + 1):
                cur[i] = min(prev[i - 1] + (a[i - 1] != b[j - 1]), prev[i] + 1, cur[i - 1] + 1)
            prev = cur
        assert T.levenshtein_naive(a, b) == prev[len(a)]
        assert T.levenshtein_naive_k_with_opts(a, b, 1000, False, T.LEVENSHTEIN_COSTS)[0] == prev[len(a)]
    with pytest.raises(NotImplementedError):                              # more than 254 distinct items COMMON to both sides
        T.levenshtein_naive(list(range(400)), list(range(400)))
    x = "".join(chr(0x400 + i) for i in range(200))
    assert T.levenstein_naive_str(x + "".join(chr(0x1000 + i) for i in range(300)), x[1:] + "".join(chr(0x2000 + i) for i in range(300))) == 301
    assert T.hamming(array.array("B", [1, 2, 3]), (ctypes.c_uint8 * 3)(1, 9, 3)) == 1
    with pytest.raises(TypeError):
        T.hamming(5, b"abcde")
    with pytest.raises(T.PanicError):                                    # src/hamming.rs:136: haystack_len / 0
        list(T.hamming_search_naive_with_opts(b"", b"abc", 2, T.SearchType.Best))
    with pytest.raises(T.PanicError):
        list(T.hamming_search_naive(b"", b"abc"))


def test_queue_of_single_pairs():
    """ta_queue_*: pairs pushed one at a time, answered by one batch pass per flush -- the answers of the single calls, in push order;
    tickets restart after a flush; an empty flush is fine; general EditCosts too."""
    import triple_accel_amd as T
    g = Dg.rng(91)
    for k, costs in [(8, (1, 1, 0, None)), (30, (1, 1, 0, 1)), (12, (2, 3, 1, None))]:
        q = T.Queue(k, T.EditCosts(*costs))
        assert q.flush() == []
        for rnd, n in enumerate((1, 300, 6000)):
            pairs = []
            for i in range(n):
                x = Dg.rand_str(g, int(g.integers(0, 200)))
                y = Dg.mutate(g, x, int(g.integers(0, 12)), costs[3] is not None) if i % 3 else Dg.rand_str(g, int(g.integers(0, 200)))
                assert q.push(x, y) == i
                pairs.append((x, y))
            got = q.flush()
            want = O.levenshtein_k_batch(O.csr_from_list([p[0] for p in pairs]), O.csr_from_list([p[1] for p in pairs]), k, costs)
            assert got == [None if int(w) == 0xFFFFFFFF else int(w) for w in want], (k, costs, n)
        q.close()
        # the queue's stream is gone now; a scratch-using call on another stream must not touch it (the library once recorded its
        # cross-stream event lazily, on the previous stream: a crash here)
        a = [Dg.rand_str(g, int(g.integers(0, 60))) for _ in range(5000)]
        b = [Dg.mutate(g, x, 3) for x in a]
        assert np.array_equal(gpu_k(a, b, 4), oracle_k(a, b, 4))


def test_many_helpers_and_failed_flush():
    """levenshtein_simd_k_with_opts_many / levenshtein_many: what a loop over the single-call function returns, through the queue (several
    flushes); and a flush that fails drops its pairs -- the queue works again afterwards (it used to fail the same way for ever)."""
    import triple_accel_amd as T
    g = Dg.rng(92)
    pairs = []
    for i in range(3000):
        x = Dg.rand_str(g, int(g.integers(0, 120)))
        pairs.append((x, Dg.mutate(g, x, int(g.integers(0, 9))) if i % 4 else Dg.rand_str(g, int(g.integers(0, 120)))))
    want = O.levenshtein_k_batch(O.csr_from_list([p[0] for p in pairs]), O.csr_from_list([p[1] for p in pairs]), 6, (1, 1, 0, None))
    got = T.levenshtein_simd_k_with_opts_many(iter(pairs), 6, flush_every=700)
    assert got == [None if int(w) == 0xFFFFFFFF else int(w) for w in want]
    assert T.levenshtein_many(pairs[:200]) == [O.levenshtein(x, y) for x, y in pairs[:200]]
    # a flush whose pass fails (TA_FAIL_PASS: a tuning switch that makes the distance pass return TA_ERR_UNSUPPORTED): the flush raises, the
    # queue drops the pairs and works again afterwards
    q = T.Queue(12, T.EditCosts(2, 3, 1, None))
    q.push(b"kitten", b"sitting")
    q.push(b"flaw", b"lawn")
    os.environ["TA_FAIL_PASS"] = "1"
    try:
        with pytest.raises(Exception):
            q.flush()
    finally:
        del os.environ["TA_FAIL_PASS"]
    assert q.flush() == []
    q.push(b"kitten", b"sitting")
    assert q.flush() == [O.levenshtein_simd_k_with_opts(b"kitten", b"sitting", 12, False, (2, 3, 1, None))[0]]
    # 40,000-byte strings under weighted costs in a queue pass: served like any other pair
    big = Dg.rand_str(g, 40_000)
    near = Dg.mutate(g, big, 4)
    q.push(big, big[::-1])
    q.push(big, near)
    assert q.flush() == [O.levenshtein_simd_k_with_opts(big, y, 12, False, (2, 3, 1, None))[0] for y in (big[::-1], near)]
    q.close()


def test_round5_paths_on_degenerate_batches():
    """This round's batch paths on degenerate shapes: empty strings (all of them, one side only), one-byte strings, k = 0, batches of 1 / 63 /
    65 / 1,024 / 1,025 pairs -- device-driven levenshtein_exp rounds, the unit-cost pre-pass, checkpoint tracebacks (CSR and fixed-length),
    hamming_search with needle = haystack and with no hit at all."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    g = Dg.rng(0xED6E)
    shapes = []
    shapes.append(([b""] * 1500, [b""] * 1500))
    shapes.append(([b""] * 1100, [Dg.rand_str(g, int(g.integers(0, 9))) for _ in range(1100)]))
    shapes.append(([Dg.rand_str(g, 1) for _ in range(1025)], [Dg.rand_str(g, 1) for _ in range(1025)]))
    for n in (1, 63, 65, 1024, 1025):
        a = [Dg.rand_str(g, int(g.integers(0, 70))) for _ in range(n)]
        shapes.append((a, [Dg.mutate(g, x, int(g.integers(0, 6)), True) if i % 3 else Dg.rand_str(g, int(g.integers(0, 70))) for i, x in enumerate(a)]))
    for a, b in shapes:
        sa, sb, ca, cb = B.Strings.from_list(a), B.Strings.from_list(b), O.csr_from_list(a), O.csr_from_list(b)
        for costs in ((1, 1, 0, None), (1, 1, 0, 1), (2, 3, 1, None)):
            got = B.levenshtein_exp_batch(sa, sb, costs).cpu().numpy().view(np.uint32)
            assert np.array_equal(got, O.levenshtein_exp_batch(ca, cb, costs)), (len(a), costs)
        try:
            T.set_option(T.OPT_UNIT_PREFILTER, True)
            for k in (0, 3, 40):
                got = B.levenshtein_k_batch(sa, sb, k, (2, 3, 1, None)).cpu().numpy().view(np.uint32)
                assert np.array_equal(got, O.levenshtein_k_batch(ca, cb, k, (2, 3, 1, None))), (len(a), k)
        finally:
            T.set_option(T.OPT_UNIT_PREFILTER, False)
        for costs in ((1, 1, 0, None), (1, 1, 0, 1)):
            for k in (0, 2, 30):
                out, ed, ne = B.levenshtein_trace_batch(sa, sb, k, costs)
                d, scripts = out.cpu().numpy().view(np.uint32), B.edits_to_lists(ed, ne)
                for i in range(0, len(a), max(1, len(a) // 60)):
                    wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, costs)
                    assert (d[i] == wd and scripts[i] == we) if wd is not None else (d[i] == 0xFFFFFFFF and scripts[i] == []), (len(a), i, k, costs)
    # fixed-length batches of one-byte and of 16-byte strings through the folded forward sweep
    for L, n in ((1, 70), (16, 1030), (17, 64)):
        am, bm = Dg.pairs_mutated_fixed(0xED70 + L, n, L, 1)
        out, ed, ne = B.levenshtein_trace_batch(B.Strings.from_fixed(am), B.Strings.from_fixed(bm), 4)
        d, scripts = out.cpu().numpy().view(np.uint32), B.edits_to_lists(ed, ne)
        for i in range(n):
            wd, we = O.levenshtein_simd_k_with_opts(am[i].tobytes(), bm[i].tobytes(), 4, True)
            assert (d[i] == wd and scripts[i] == we) if wd is not None else (d[i] == 0xFFFFFFFF and scripts[i] == []), (L, i)
    # hamming_search: the haystack IS the needle; no hit anywhere; a needle longer than the haystack
    for nl in (8, 24, 33, 64, 200):
        needle = bytes(int(c) or 1 for c in Dg.random_bytes(g, nl))
        assert [tuple(int(v) for v in r) for r in B.hamming_search_dev(needle, B.haystack_tensor(needle), 0)] == [(0, nl, 0)]
        hay = bytes([7]) * 5000
        assert len(B.hamming_search_dev(bytes([9]) * nl, B.haystack_tensor(hay), nl // 4)) == 0
        assert len(B.hamming_search_dev(needle, B.haystack_tensor(needle[:-1]), 1)) == 0
