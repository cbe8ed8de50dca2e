"""-m gpu: adversarial alphabets through the driver-run suite: a fixed-seed slice of scripts/fuzz.py and
scripts/r05/fuzz_trace.py.  NUL is what the kernels pad with outside the strings and what the reference's SIMD windows are padded with
(tests/basic_tests.rs:503-537, 774-802 are the reference's own NUL cases); 0x0C is the byte-test constant of the bit-parallel kernels
(`x ^ 0x0C0C0C0C` + v_perm, wave.h ne12).  Alphabets {0x00}, {0x00, 0x0C}, {0x0C, 0x0D}, 0..255 through: the checkpoint trace kernel (fixed,
CSR, length-ordered routes, every TILE / STILE), the packed record form, the weighted search filter, the device-driven exp rounds and the
unit pre-pass -- each against the oracle, bit for bit / edit for edit.
Mutation check (done once, by hand, r06): with DevWave::ne12 comparing against 13 instead of 12 this file fails in every test; the zeroing
of the string pieces outside a string in lev_bits_trace_body.h:load_strings is NOT load-bearing (row / column masks decide): removing it keeps
every test green -- in the emulation too (tests/test_emu_lev_bits_trace.py)."""
import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O

pytestmark = pytest.mark.gpu

ALPHABETS = {"nul": [0], "nul_0c": [0, 0x0C], "0c_0d": [0x0C, 0x0D], "all": list(range(256))}


def _mutate(g, x, sym, edits, trans):
    y = bytearray(x)
    for _ in range(edits):
        t = int(g.integers(0, 4 if trans else 3))
        pos = int(g.integers(0, len(y) + 1))
        c = int(sym[int(g.integers(0, len(sym)))])
        if t == 0 and pos < len(y):
            y[pos] = c
        elif t == 1:
            y.insert(pos, c)
        elif t == 2 and pos < len(y):
            del y[pos]
        elif t == 3 and pos + 1 < len(y):
            y[pos], y[pos + 1] = y[pos + 1], y[pos]
    return bytes(y)


def _ragged(g, name, n_pairs, max_len, k, trans):
    sym = np.array(ALPHABETS[name], dtype=np.uint8)
    a, b = [], []
    for i in range(n_pairs):
        x = bytes(sym[g.integers(0, len(sym), size=int(g.integers(0, max_len)))])
        y = _mutate(g, x, sym, int(g.integers(0, k + 2)), trans) if i % 6 else bytes(sym[g.integers(0, len(sym), size=int(g.integers(0, max_len)))])
        if i % 2:
            x, y = y, x
        a.append(x); b.append(y)
    return a, b


def _fixed(g, name, n_pairs, L, k, trans):
    sym = np.array(ALPHABETS[name], dtype=np.uint8)
    a = sym[g.integers(0, len(sym), size=(n_pairs, L))]
    b = a.copy()
    for i in range(n_pairs):
        m = (_mutate(g, a[i].tobytes(), sym, int(g.integers(0, k // 2 + 2)), trans) + bytes(sym[g.integers(0, len(sym), size=L)]))[:L]
        b[i] = np.frombuffer(m, dtype=np.uint8)
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def _check_scripts(a, b, k, costs, got_d, got_e, step=1):
    some = 0
    for i in range(0, len(a), step):
        wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, costs)
        if wd is None:
            assert got_d[i] == 0xFFFFFFFF and got_e[i] == [], i
        else:
            some += 1
            assert got_d[i] == wd and got_e[i] == we, (i, a[i], b[i], got_e[i], we)
    return some


@pytest.mark.parametrize("name", sorted(ALPHABETS))
@pytest.mark.parametrize("trans", [False, True])
def test_checkpoint_trace_kernel_adversarial_alphabets(name, trans, monkeypatch):
    """CSR (small: batch order; 4,500 pairs: length-ordered), fixed-length (folded sweep) and the kernel's own sweep with every TILE / STILE."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    costs = (1, 1, 0, 1) if trans else (1, 1, 0, None)
    g = Dg.rng(0xADA + len(name) + int(trans))
    k = 30 if trans else 32
    a, b = _ragged(g, name, 4500, 200, k, trans)
    sa, sb = B.Strings.from_list(a), B.Strings.from_list(b)
    out, edits, ne = B.levenshtein_trace_batch(sa, sb, k, costs)
    assert T.last_launch_info()["kernel"] == 8
    d, e = out.cpu().numpy().view(np.uint32), B.edits_to_lists(edits, ne)
    assert _check_scripts(a, b, k, costs, d, e, step=5) > 300
    # the packed form of the same batch: the same scripts
    outp, packed, nep = B.levenshtein_trace_batch_packed(sa, sb, k, costs)
    assert np.array_equal(outp.cpu().numpy().view(np.uint32), d) and B.packed_to_lists(packed, nep) == e
    for tile, stile in ((8, 32), (8, 64), (16, 32), (32, 64)):
        monkeypatch.setenv("TA_TRACE_TILE", str(tile)); monkeypatch.setenv("TA_TRACE_STILE", str(stile))
        o2, e2, n2 = B.levenshtein_trace_batch(B.Strings.from_list(a[:700]), B.Strings.from_list(b[:700]), 9, costs)
        monkeypatch.delenv("TA_TRACE_TILE"); monkeypatch.delenv("TA_TRACE_STILE")
        assert _check_scripts(a[:700], b[:700], 9, costs, o2.cpu().numpy().view(np.uint32), B.edits_to_lists(e2, n2), step=3) > 40
    fa, fb = _fixed(g, name, 1500, 150, k, trans)
    for xa, xb in ((fa, fb), (np.ascontiguousarray(fa[:, :140]), fb)):
        o3, e3, n3 = B.levenshtein_trace_batch(B.Strings.from_fixed(xa), B.Strings.from_fixed(xb), k, costs)
        assert T.last_kernel_name().endswith("true>")
        la, lb = [r.tobytes() for r in xa], [r.tobytes() for r in xb]
        assert _check_scripts(la, lb, k, costs, o3.cpu().numpy().view(np.uint32), B.edits_to_lists(e3, n3), step=4) > 150


@pytest.mark.parametrize("name", sorted(ALPHABETS))
def test_distance_routes_adversarial_alphabets(name, monkeypatch):
    """k-bounded batches (bit-parallel band, two pairs per lane, DP band), the device-driven exp rounds and the unit pre-pass."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    g = Dg.rng(0xADB + len(name))
    for L, k, costs in ((256, 32, (1, 1, 0, None)), (128, 8, (1, 1, 0, 1)), (96, 20, (2, 3, 1, None)), (96, 12, (2, 2, 1, 3))):
        fa, fb = _fixed(g, name, 3000, L, max(2, k // max(costs[0], costs[1])), costs[3] is not None)
        got = B.levenshtein_k_batch(B.Strings.from_fixed(fa), B.Strings.from_fixed(fb), k, costs).cpu().numpy().view(np.uint32)
        want = O.levenshtein_k_batch(O.csr_from_fixed(fa), O.csr_from_fixed(fb), k, costs)
        assert np.array_equal(got, want), (name, L, k, costs)
    a, b = _ragged(g, name, 6000, 300, 40, True)
    sa, sb = B.Strings.from_list(a), B.Strings.from_list(b)
    ca, cb = O.csr_from_list(a), O.csr_from_list(b)
    for costs in ((1, 1, 0, None), (1, 1, 0, 1)):
        got = B.levenshtein_exp_batch(sa, sb, costs).cpu().numpy().view(np.uint32)              # >= 1,024 pairs: device-driven rounds
        assert np.array_equal(got, O.levenshtein_exp_batch(ca, cb, costs)), (name, costs)
    T.set_option(T.OPT_UNIT_PREFILTER, True)
    try:
        for k, costs in ((30, (2, 3, 1, None)), (24, (3, 2, 0, 3))):
            got = B.levenshtein_k_batch(sa, sb, k, costs).cpu().numpy().view(np.uint32)
            assert np.array_equal(got, O.levenshtein_k_batch(ca, cb, k, costs)), (name, k, costs)
    finally:
        T.set_option(T.OPT_UNIT_PREFILTER, False)


@pytest.mark.parametrize("name", sorted(ALPHABETS))
def test_search_filter_adversarial_alphabets(name):
    """levenshtein_search through the unit-cost scan (as the exact filter and as the weighted superset filter) and hamming_search's naive
    contract (NUL bytes are data there)."""
    import triple_accel_amd as T
    sym = np.array(ALPHABETS[name], dtype=np.uint8)
    g = Dg.rng(0xADC + len(name))
    hay = bytearray(sym[g.integers(0, len(sym), size=150_000)].tobytes())
    needle = bytes(sym[g.integers(0, len(sym), size=20)]) if len(sym) > 2 else bytes([sym[0]] * 12 + [sym[-1]] * 3 + [sym[0]] * 5)
    for pos in range(5000, len(hay) - 100, 9000):
        m = _mutate(g, needle, sym, 3, True)
        hay[pos:pos + len(m)] = m
    hay = bytes(hay)
    if len(sym) <= 2:                     # a two-letter haystack matches almost everywhere: keep the result lists small
        hay = hay[:20_000]
    for k, costs in ((3, (1, 1, 0, None)), (3, (1, 1, 0, 1)), (5, (2, 1, 1, None)), (6, (2, 3, 1, None))):
        for st in (T.SearchType.All, T.SearchType.Best):
            got = [tuple(m) for m in T.levenshtein_search_simd_with_opts(needle, hay, k, st, T.EditCosts(*costs), False)]
            assert got == O.levenshtein_search_naive_with_opts(needle, hay, k, st, costs, False), (name, k, costs, st)
    got = [tuple(m) for m in T.hamming_search_naive_with_opts(needle, hay, 4, T.SearchType.All)]
    assert got == O.hamming_search_naive_with_opts(needle, hay, 4, O.ALL)


@pytest.mark.parametrize("costs,k", [((2, 2, 0, None), 64), ((3, 3, 0, 3), 60), ((2, 2, 0, None), 7)])
def test_trace_batch_unit_costs_times_g(costs, k, monkeypatch):
    """EditCosts(g, g, 0, None | Some(g)): the checkpoint kernel with k / g -- the oracle's scripts under the caller's costs, the distances
    times g -- and the DP band kernel's records (TA_NO_UNIT_SCALE=1) agree."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    g = Dg.rng(0x6E6 + k)
    am, bm = Dg.pairs_mutated_fixed(0x6E7 + k, 2500, 180, 20 if k > 10 else 3, swaps=costs[3] is not None)
    sa, sb = B.Strings.from_fixed(am), B.Strings.from_fixed(bm)
    out, edits, ne = B.levenshtein_trace_batch(sa, sb, k, costs)
    assert T.last_launch_info()["kernel"] == 8, T.last_launch_info()
    d, e = out.cpu().numpy().view(np.uint32), B.edits_to_lists(edits, ne)
    la, lb = [r.tobytes() for r in am], [r.tobytes() for r in bm]
    assert _check_scripts(la, lb, k, costs, d, e, step=3) > 300
    monkeypatch.setenv("TA_NO_UNIT_SCALE", "1")
    o2, e2, n2 = B.levenshtein_trace_batch(sa, sb, k, costs)
    monkeypatch.delenv("TA_NO_UNIT_SCALE")
    assert T.last_launch_info()["kernel"] == 1
    assert np.array_equal(d, o2.cpu().numpy().view(np.uint32)) and e == B.edits_to_lists(e2, n2)
    a, b = _ragged(g, "all", 3000, 150, k // costs[0], costs[3] is not None)
    o3, p3, n3 = B.levenshtein_trace_batch_packed(B.Strings.from_list(a), B.Strings.from_list(b), k, costs)
    assert _check_scripts(a, b, k, costs, o3.cpu().numpy().view(np.uint32), B.packed_to_lists(p3, n3), step=4) > 100


def test_trace_batch_packed_on_the_record_routes_and_cut_scripts():
    """The packed form where the walk writes ta_edit records (weighted costs: the DP band kernel) -- packed on the device from a bounded
    scratch -- and scripts longer than the slot: the LAST cap runs, n_edits = the true length, nothing written in front."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    g = Dg.rng(0x9AC)
    a, b = _ragged(g, "all", 1500, 120, 12, False)
    sa, sb = B.Strings.from_list(a), B.Strings.from_list(b)
    for k, costs in ((20, (2, 3, 1, None)), (18, (2, 2, 1, 3))):
        o, p, n = B.levenshtein_trace_batch_packed(sa, sb, k, costs)
        assert T.last_launch_info()["kernel"] == 1
        assert _check_scripts(a, b, k, costs, o.cpu().numpy().view(np.uint32), B.packed_to_lists(p, n), step=2) > 100
    for costs in ((1, 1, 0, None), (2, 3, 0, None)):
        k = 12 * costs[0]
        import torch
        packed = torch.full((len(a), 5), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        o, p, n = B.levenshtein_trace_batch_packed(sa, sb, k, costs, cap=5, packed=packed)
        pl, nn, raw = B.packed_to_lists(p, n, allow_cut=True), n.cpu().numpy(), p.cpu().numpy()
        cut = 0
        for i in range(0, len(a), 2):
            wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, costs)
            if wd is None:
                assert nn[i] == 0 and np.all(raw[i] == 0x5A5A5A5A)
                continue
            assert nn[i] == len(we) and pl[i] == we[max(0, len(we) - 5):], (i, costs)
            assert np.all(raw[i, :5 - min(len(we), 5)] == 0x5A5A5A5A)
            cut += len(we) > 5
        assert cut > 20


def test_trace_batch_in_chunks(monkeypatch):
    """the checkpoint route indexes its run lists with 32 bits and sized its scratch for the whole batch.  Sub-batches of whole
    wavefronts (TA_TRACE_CHUNK_PAIRS pins a small chunk here) give the same distances and scripts -- fixed-length and CSR, both record forms."""
    from triple_accel_amd import batch as B
    am, bm = Dg.pairs_mutated_fixed(0xC4C, 1000, 120, 10)
    g = Dg.rng(0xC4D)
    a, b = _ragged(g, "all", 1000, 130, 10, False)
    for sa, sb in ((B.Strings.from_fixed(am), B.Strings.from_fixed(bm)), (B.Strings.from_list(a), B.Strings.from_list(b))):
        o, e, n = B.levenshtein_trace_batch(sa, sb, 14)
        op, pp, npk = B.levenshtein_trace_batch_packed(sa, sb, 14)
        monkeypatch.setenv("TA_TRACE_CHUNK_PAIRS", "192")
        o2, e2, n2 = B.levenshtein_trace_batch(sa, sb, 14)
        o3, p3, n3 = B.levenshtein_trace_batch_packed(sa, sb, 14)
        monkeypatch.delenv("TA_TRACE_CHUNK_PAIRS")
        assert np.array_equal(o.cpu().numpy(), o2.cpu().numpy()) and B.edits_to_lists(e, n) == B.edits_to_lists(e2, n2)
        assert np.array_equal(o.cpu().numpy(), o3.cpu().numpy()) and B.packed_to_lists(pp, npk) == B.packed_to_lists(p3, n3) == B.edits_to_lists(e, n)
    assert _check_scripts(a, b, 14, (1, 1, 0, None), o2.cpu().numpy().view(np.uint32), B.edits_to_lists(e2, n2), step=7) > 50      # (the CSR batch, chunked)


def test_captured_call_refuses_to_grow_its_scratch():
    """a captured call bakes the thread's scratch pointers into the graph; growing the scratch inside a capture is illegal.  The
    library now says so (TA_ERR_UNSUPPORTED) instead of calling hipFree / hipMalloc under the capture."""
    import torch
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    T.thread_release()                                        # no scratch held
    a, b = _ragged(Dg.rng(7), "all", 5000, 100, 8, False)
    sa, sb = B.Strings.from_list(a), B.Strings.from_list(b)
    out = torch.empty(len(a), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(NotImplementedError):
        with torch.cuda.graph(graph, stream=side):
            B.levenshtein_k_batch(sa, sb, 8, out=out)         # (a CSR batch of >= 4,096 pairs orders its pairs in scratch)
    torch.cuda.synchronize()
    got = B.levenshtein_k_batch(sa, sb, 8, out=out)           # outside a capture the same call sizes the scratch ...
    torch.cuda.synchronize()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=side):               # ... and is then capturable
        B.levenshtein_k_batch(sa, sb, 8, out=out)
    out.zero_()
    graph2.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), O.levenshtein_k_batch(O.csr_from_list(a), O.csr_from_list(b), 8))
