"""-m gpu: traceback (trace_on = true) -- 2-bit argmin codes from the band-wavefront kernel, walk on the host --
against the oracle's scalar traceback (src/levenshtein.rs:561-606), edit for edit."""
import numpy as np
import pytest

import datagen as Dg
import oracle_lib as O

pytestmark = pytest.mark.gpu

COSTS = [(1, 1, 0, None), (1, 1, 0, 1), (2, 3, 0, None), (3, 1, 0, None), (1, 1, 2, None), (2, 1, 2, None), (2, 2, 1, 3)]


def prod(a, b, k, costs):
    import triple_accel_amd as T
    r = T.levenshtein_simd_k_with_opts(a, b, k, True, T.EditCosts(*costs))
    return (None, None) if r is None else (r[0], [tuple(e) for e in r[1]])


@pytest.mark.parametrize("costs", COSTS)
def test_trace_equals_scalar(costs):
    g = Dg.rng(31)
    for it in range(120):
        a = Dg.rand_str(g, int(g.integers(0, 40)))
        b = Dg.mutate(g, a, 6, costs[3] is not None) if it % 3 else Dg.rand_str(g, int(g.integers(0, 40)))
        for k in (2, 7, 30, 0xFFFFFFFF):
            want = O.levenshtein_simd_k_with_opts(a, b, k, True, costs)
            if want[0] is None:
                want = (None, None)
            assert prod(a, b, k, costs) == want, (a, b, k, costs)


def test_trace_small_alphabet_ties_and_swap():
    """Binary alphabets maximise ties; a longer first argument exercises the swap + AGap/BGap relabelling."""
    g = Dg.rng(32)
    for costs in COSTS:
        for _ in range(80):
            a = g.integers(97, 99, size=int(g.integers(0, 14)), dtype=np.uint8).tobytes()
            b = g.integers(97, 99, size=int(g.integers(0, 14)), dtype=np.uint8).tobytes()
            want = O.levenshtein_simd_k_with_opts(a, b, 0xFFFFFFFF, True, costs)
            assert prod(a, b, 0xFFFFFFFF, costs) == want, (a, b, costs)


def test_trace_longer_strings_and_exp():
    import triple_accel_amd as T
    g = Dg.rng(33)
    for n in (100, 300, 900):
        a = Dg.rand_str(g, n)
        b = Dg.mutate(g, a, n // 8, True)
        for costs in [(1, 1, 0, None), (1, 1, 0, 1), (1, 1, 2, None)]:
            want = O.levenshtein_simd_k_with_opts(a, b, n, True, costs)
            assert prod(a, b, n, costs) == want
        d, tr = T.levenshtein_exp_with_opts(a, b, True, T.LEVENSHTEIN_COSTS)
        wd, wtr = O.levenshtein_exp_with_opts(a, b, True)
        assert (d, [tuple(e) for e in tr]) == (wd, wtr)


def test_trace_beyond_the_register_band():
    """Unit-cost tracebacks whose band needs more than 64 x 66 diagonals take the row-blocked bit-parallel kernel with
    3-bit records and the host walk: same edit script as the scalar path, for several stripes, swapped roles and the
    transposition family; weighted costs on such bands stay unsupported."""
    import triple_accel_amd as T
    g = Dg.rng(34)
    for n in (5000, 9000):
        a = Dg.rand_str(g, n)
        b = Dg.mutate(g, a, n // 20, True)
        for costs in [(1, 1, 0, None), (1, 1, 0, 1)]:
            for x, y in ((a, b), (b, a)):
                want = O.levenshtein_simd_k_with_opts(x, y, 0xFFFFFFFF, True, costs)
                assert prod(x, y, 0xFFFFFFFF, costs) == want, (n, costs)
        d, tr = T.levenshtein_exp_with_opts(a, b, True, T.RDAMERAU_COSTS)
        wd, wtr = O.levenshtein_exp_with_opts(a, b, True, O.RDAMERAU_COSTS)
        assert (d, [tuple(e) for e in tr]) == (wd, wtr)
    s1 = g.integers(97, 100, size=6000, dtype=np.uint8).tobytes()
    s2 = g.integers(97, 100, size=5500, dtype=np.uint8).tobytes()
    assert prod(s1, s2, 0xFFFFFFFF, (1, 1, 0, 1)) == O.levenshtein_simd_k_with_opts(s1, s2, 0xFFFFFFFF, True, (1, 1, 0, 1))
    # weighted / affine / transposition costs on such bands: the DP wide kernel with 2-bit argmin codes
    a = Dg.rand_str(g, 4700)
    b = Dg.mutate(g, a, 120, True)
    for costs in [(2, 1, 0, None), (1, 1, 1, None), (3, 2, 1, 3), (2, 2, 0, 2)]:
        for x, y in ((a, b), (b, a)):
            want = O.levenshtein_simd_k_with_opts(x, y, 0xFFFFFFFF, True, costs)
            assert prod(x, y, 0xFFFFFFFF, costs) == want, costs
    assert prod(s1, s2, 0xFFFFFFFF, (2, 2, 0, 2)) == O.levenshtein_simd_k_with_opts(s1, s2, 0xFFFFFFFF, True, (2, 2, 0, 2))


@pytest.mark.parametrize("costs,k", [((1, 1, 0, None), 32), ((1, 1, 0, 1), 12), ((2, 3, 1, None), 40), ((2, 2, 1, 3), 20), ((3, 1, 0, None), 25)])
def test_trace_batch_equals_scalar(costs, k):
    """ta_levenshtein_trace_batch: distances and edit scripts of a whole ragged batch from the device (argmin codes + the walk kernel),
    edit for edit the oracle's scalar traceback -- near pairs, unrelated pairs (None), swapped roles (a longer than b), empty strings,
    small alphabets (ties); n_edits = 0 exactly where the distance is None."""
    from triple_accel_amd import batch as B
    g = Dg.rng(41 + k)
    a, b = [], []
    for i in range(3000):
        t = i % 6
        if t == 5:
            x = g.integers(97, 99, size=int(g.integers(0, 30)), dtype=np.uint8).tobytes()
            y = g.integers(97, 99, size=int(g.integers(0, 30)), dtype=np.uint8).tobytes()
        else:
            x = Dg.rand_str(g, int(g.integers(0, 200)))
            y = Dg.rand_str(g, int(g.integers(0, 200))) if t == 0 else Dg.mutate(g, x, 9, costs[3] is not None)
            if t == 1:
                x, y = y, x
        a.append(x); b.append(y)
    out, edits, ne = B.levenshtein_trace_batch(B.Strings.from_list(a), B.Strings.from_list(b), k, costs)
    got_d = out.cpu().numpy().view(np.uint32)
    got_e = B.edits_to_lists(edits, ne)
    n_some = 0
    for i in range(len(a)):
        wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, costs)
        if wd is None:
            assert got_d[i] == 0xFFFFFFFF and got_e[i] == [], i
        else:
            n_some += 1
            assert got_d[i] == wd and got_e[i] == we, (i, a[i], b[i], got_e[i], we)
    assert n_some > 1000


def test_trace_batch_fixed_length_cfg2_shape_and_cap():
    """cfg2's geometry (256-byte strings, k = 32, mutated pairs), a fixed-length batch large enough for several chunks of records is not
    needed here -- 20,000 pairs; plus the cap: a script longer than `cap` runs is cut and n_edits says how long it is."""
    from triple_accel_amd import batch as B
    am, bm = Dg.pairs_mutated_fixed(0x7AA2, 20_000, 256, 24)
    out, edits, ne = B.levenshtein_trace_batch(B.Strings.from_fixed(am), B.Strings.from_fixed(bm), 32)
    d = out.cpu().numpy().view(np.uint32)
    want_d = O.levenshtein_k_batch(O.csr_from_fixed(am), O.csr_from_fixed(bm), 32)
    assert np.array_equal(d, want_d)
    got = B.edits_to_lists(edits, ne)
    for i in range(0, 20_000, 97):
        wd, we = O.levenshtein_simd_k_with_opts(am[i].tobytes(), bm[i].tobytes(), 32, True)
        assert (got[i] == we) if wd is not None else (got[i] == []), i
    out2, edits2, ne2 = B.levenshtein_trace_batch(B.Strings.from_fixed(am[:500]), B.Strings.from_fixed(bm[:500]), 32, cap=3)
    assert np.array_equal(ne2.cpu().numpy(), ne[:500].cpu().numpy())
    got2 = B.edits_to_lists(edits2, ne2, allow_cut=True)
    for i in range(500):
        assert got2[i] == got[i][:3], i
    with pytest.raises(ValueError):
        B.edits_to_lists(edits2, ne2)                       # a cut script is an error unless asked for


def test_trace_batch_csr_side_without_max_len():
    """A CSR side built as Strings(blob, off) carries max_len = 0 ("let the library measure it", triple_accel_amd.h): the default
    cap must come from the offsets, not from a length of 0 (it was min(2k+1, 3) = 3 and every longer script was cut
    without an error)."""
    from triple_accel_amd import batch as B
    g = Dg.rng(0xC5A)
    a = [Dg.rand_str(g, int(g.integers(20, 120))) for _ in range(400)]
    b = [Dg.mutate(g, x, 7, False) for x in a]
    sa, sb = B.Strings.from_list(a), B.Strings.from_list(b)
    ra, rb = B.Strings(sa.blob, sa.off), B.Strings(sb.blob, sb.off)        # no max_len
    assert ra.max_len == 0
    out, edits, ne = B.levenshtein_trace_batch(ra, rb, 16)
    assert edits.shape[1] == 33
    got = B.edits_to_lists(edits, ne)
    d = out.cpu().numpy().view(np.uint32)
    for i in range(len(a)):
        wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], 16, True)
        assert (d[i] == wd and got[i] == we) if wd is not None else (d[i] == 0xFFFFFFFF and got[i] == []), i


@pytest.mark.parametrize("trans,k", [(False, 32), (True, 30), (False, 5), (True, 9)])
def test_trace_batch_checkpoint_kernel(trans, k, monkeypatch):
    """The unit-cost families with a band of up to 33 diagonals take the checkpoint-and-recompute kernel (lev_bits_trace_body.h: no per-cell
    records): its scripts are the oracle's and the DP band kernel's (TA_TRACE_NO_BITS=1), edit for edit -- ragged batches with strings of
    up to 1,500 bytes (dozens of tiles), both orientations, ties on a binary alphabet, None pairs inside a wavefront, tiles of 16 and 32
    columns, fixed-length batches."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    costs = (1, 1, 0, 1) if trans else (1, 1, 0, None)
    g = Dg.rng(0xCB7 + k + int(trans))
    a, b = [], []
    for i in range(2000):
        t = i % 7
        if t == 6:
            x = g.integers(97, 99, size=int(g.integers(0, 60)), dtype=np.uint8).tobytes()
            y = g.integers(97, 99, size=int(g.integers(0, 60)), dtype=np.uint8).tobytes()
        else:
            n = int(g.integers(0, 1500 if i % 50 == 0 else 260))
            x = Dg.rand_str(g, n)
            y = Dg.rand_str(g, int(g.integers(0, 260))) if t == 0 else Dg.mutate(g, x, int(g.integers(0, k + 2)), trans)
            if t in (1, 2):
                x, y = y, x
        a.append(x); b.append(y)
    sa, sb = B.Strings.from_list(a), B.Strings.from_list(b)
    out, edits, ne = B.levenshtein_trace_batch(sa, sb, k, costs)
    assert "lev_bits_trace_kernel" in T.last_kernel_name() and T.last_launch_info()["kernel"] == 8
    got_d, got_e = out.cpu().numpy().view(np.uint32), B.edits_to_lists(edits, ne)
    monkeypatch.setenv("TA_TRACE_TILE", "32")
    out32, edits32, ne32 = B.levenshtein_trace_batch(sa, sb, k, costs)
    assert ", 32, " in T.last_kernel_name()
    monkeypatch.delenv("TA_TRACE_TILE")
    monkeypatch.setenv("TA_TRACE_NO_BITS", "1")
    out_dp, edits_dp, ne_dp = B.levenshtein_trace_batch(sa, sb, k, costs)
    assert "lev_band_trace_kernel" in T.last_kernel_name()
    monkeypatch.delenv("TA_TRACE_NO_BITS")
    assert np.array_equal(got_d, out_dp.cpu().numpy().view(np.uint32)) and np.array_equal(got_d, out32.cpu().numpy().view(np.uint32))
    assert got_e == B.edits_to_lists(edits_dp, ne_dp) and got_e == B.edits_to_lists(edits32, ne32)
    n_some = 0
    for i in range(0, len(a), 3):
        wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, costs)
        if wd is None:
            assert got_d[i] == 0xFFFFFFFF and got_e[i] == [], i
        else:
            n_some += 1
            assert got_d[i] == wd and got_e[i] == we, (i, a[i], b[i], got_e[i], we)
    assert n_some > 300
    # fixed-length, a longer than b and b longer than a
    am, bm = Dg.pairs_mutated_fixed(0xCB8 + k, 3000, 200, max(2, k // 2), swaps=trans)
    # (fixed-length batches: the distance pass is the forward sweep -- the stride-8 kernel's CKPT instantiation leaves the checkpoints,
    # the trace kernel's HAVE_CKPT instantiation starts with the walk; TA_TRACE_OWN_SWEEP=1: the trace kernel's own sweep, an A/B)
    for xa, xb in ((am, np.ascontiguousarray(bm[:, :197])), (np.ascontiguousarray(am[:, :195]), bm), (np.ascontiguousarray(am[:, :96]), np.ascontiguousarray(bm[:, :100]))):
        o2, e2, n2 = B.levenshtein_trace_batch(B.Strings.from_fixed(xa), B.Strings.from_fixed(xb), k, costs)
        assert "lev_bits_trace_kernel" in T.last_kernel_name() and T.last_kernel_name().endswith("true>"), T.last_kernel_name()
        d2, l2 = o2.cpu().numpy().view(np.uint32), B.edits_to_lists(e2, n2)
        monkeypatch.setenv("TA_TRACE_OWN_SWEEP", "1")
        o3, e3, n3 = B.levenshtein_trace_batch(B.Strings.from_fixed(xa), B.Strings.from_fixed(xb), k, costs)
        assert not T.last_kernel_name().endswith("true>")
        monkeypatch.delenv("TA_TRACE_OWN_SWEEP")
        assert np.array_equal(d2, o3.cpu().numpy().view(np.uint32)) and l2 == B.edits_to_lists(e3, n3)
        for i in range(0, 3000, 41):
            wd, we = O.levenshtein_simd_k_with_opts(xa[i].tobytes(), xb[i].tobytes(), k, True, costs)
            assert (d2[i] == wd and l2[i] == we) if wd is not None else (d2[i] == 0xFFFFFFFF and l2[i] == []), i


@pytest.mark.parametrize("trans,k", [(False, 32), (True, 12)])
def test_trace_batch_csr_in_length_order(trans, k, monkeypatch):
    """CSR batches: the distance pass is the forward sweep here too (rows = the shorter string pair by pair inside the kernel), and from
    4,096 pairs on the distance pass and the trace kernel take the pairs in length order (one list for both): the same
    distances and scripts as in batch order (TA_NO_LENGTH_ORDER=1), a sample against the oracle -- lengths 0..300, both orientations,
    None pairs and pairs outside the band anywhere in the order."""
    import triple_accel_amd as T
    from triple_accel_amd import batch as B
    costs = (1, 1, 0, 1) if trans else (1, 1, 0, None)
    g = Dg.rng(0x5EED + k)
    a, b = [], []
    for i in range(9000):
        x = Dg.rand_str(g, int(g.integers(0, 300)))
        t = i % 9
        y = Dg.rand_str(g, int(g.integers(0, 300))) if t == 0 else Dg.mutate(g, x, int(g.integers(0, k + 3)), trans)
        if t in (1, 2, 3):
            x, y = y, x
        a.append(x); b.append(y)
    sa, sb = B.Strings.from_list(a), B.Strings.from_list(b)
    out, edits, ne = B.levenshtein_trace_batch(sa, sb, k, costs)
    assert "lev_bits_trace_kernel" in T.last_kernel_name()
    got_d, got_e = out.cpu().numpy().view(np.uint32), B.edits_to_lists(edits, ne)
    assert T.last_kernel_name().endswith("true>"), T.last_kernel_name()      # (the distance pass was the forward sweep: HAVE_CKPT)
    monkeypatch.setenv("TA_NO_LENGTH_ORDER", "1")
    out1, edits1, ne1 = B.levenshtein_trace_batch(sa, sb, k, costs)
    monkeypatch.delenv("TA_NO_LENGTH_ORDER")
    assert np.array_equal(got_d, out1.cpu().numpy().view(np.uint32)) and got_e == B.edits_to_lists(edits1, ne1)
    monkeypatch.setenv("TA_TRACE_CSR_OWN_SWEEP", "1")                         # the trace kernel's own forward sweep: an A/B route
    out2, edits2, ne2 = B.levenshtein_trace_batch(sa, sb, k, costs)
    assert not T.last_kernel_name().endswith("true>")
    monkeypatch.delenv("TA_TRACE_CSR_OWN_SWEEP")
    assert np.array_equal(got_d, out2.cpu().numpy().view(np.uint32)) and got_e == B.edits_to_lists(edits2, ne2)
    n_some = 0
    for i in range(0, len(a), 7):
        wd, we = O.levenshtein_simd_k_with_opts(a[i], b[i], k, True, costs)
        if wd is None:
            assert got_d[i] == 0xFFFFFFFF and got_e[i] == [], i
        else:
            n_some += 1
            assert got_d[i] == wd and got_e[i] == we, (i, a[i], b[i], got_e[i], we)
    assert n_some > 400


def test_trace_batch_in_a_captured_graph():
    """ta_levenshtein_trace_batch synchronises nothing and fills nothing with memset nodes: the call (distance pass with checkpoints + the
    trace kernel; the DP route's two kernels for weighted costs) can be captured into a graph and replayed -- same scripts."""
    import torch
    from triple_accel_amd import batch as B
    am, bm = Dg.pairs_mutated_fixed(0x6A0, 5000, 180, 14, swaps=True)
    sa, sb = B.Strings.from_fixed(am), B.Strings.from_fixed(bm)
    for costs, k in (((1, 1, 0, 1), 30), ((2, 3, 1, None), 40)):
        out, ed, ne = B.levenshtein_trace_batch(sa, sb, k, costs)                      # (sizes the library's scratch)
        want_d, want_s = out.cpu().numpy().copy(), B.edits_to_lists(ed, ne)
        torch.cuda.synchronize()
        st, gr = torch.cuda.Stream(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(gr, stream=st):
                B.levenshtein_trace_batch(sa, sb, k, costs, out=out, edits=ed, n_edits=ne)
        out.fill_(7); ne.fill_(0); ed.fill_(0)
        gr.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want_d) and B.edits_to_lists(ed, ne) == want_s, costs
    wd, we = O.levenshtein_simd_k_with_opts(am[5].tobytes(), bm[5].tobytes(), 40, True, (2, 3, 1, None))
    assert (want_s[5] == we) if wd is not None else (want_s[5] == [])
