"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).

The hot path shards without any data-path collective:
  * pair batches: contiguous N/G pairs per rank; results stay on the rank (or are all-gathered on request):
    levenshtein_k_batch_sharded, levenshtein_exp_batch_sharded, hamming_batch_sharded, levenshtein_trace_batch_sharded;
  * hamming_search over one big haystack: the windows that START in a rank's shard need needle_len - 1 bytes of the
    ranks behind it -- every rank publishes its first needle_len - 1 bytes, searches its shard in place and a tail buffer of at most
    2 (needle_len - 1) bytes; one match-list gather, the Best pass (no overlap fold) on every rank: hamming_search_sharded;
  * levenshtein_search over one big haystack: contiguous shards RESIDENT in HBM (never copied or re-uploaded); every
    rank publishes its last needle_len + unit_k + 2 bytes (one all-gather of that many bytes, on the device under RCCL)
    and searches [left context | its first halo bytes] as a tiny head buffer plus its shard in place; every rank emits
    All-mode hits for its own end positions only; the ONE exchange step of the data path is the final match-list gather
    (counts, then padded records), after which the order-dependent Best fold runs identically on every rank.
"""
import ctypes as _C

import numpy as np
import torch
import torch.distributed as dist

from . import _native as _n
from . import EditCosts, LEVENSHTEIN_COSTS, Match, SearchType, _costs


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n independent units owned by `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _device_for_group(group=None):
    backend = dist.get_backend(group)
    return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")


def all_gather_results(local, group=None):
    """All-gather per-rank result vectors of possibly different lengths (pair batches)."""
    world = dist.get_world_size(group)
    dev = _device_for_group(group)
    local = local.to(dev)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    mx = int(max(int(c.item()) for c in counts))
    pad = torch.zeros(mx, dtype=local.dtype, device=dev)
    pad[: local.numel()] = local
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[: int(c.item())] for o, c in zip(outs, counts)])


def levenshtein_k_batch_sharded(a, b, k, costs=LEVENSHTEIN_COSTS, gather=True, group=None, local_batch=None):
    """levenshtein_simd_k_with_opts over a pair batch sharded across the ranks (contiguous pairs per rank, NO data-path
    collective).  `a`, `b`: the WHOLE batch as lists of bytes (the same on every rank: each takes its shard_range slice), or this rank's
    own slice as batch.Strings already resident in HBM.  Returns int32 distances (-1 == None): of the whole batch in pair order on every
    rank (`gather`: one ragged all-gather of 4 bytes per pair, the only exchange), or of the rank's slice (gather=False: the results stay
    where they were computed).  `local_batch(a_list, b_list, k, costs) -> int32 array` replaces the HIP path (tests, gloo)."""
    costs = _costs(costs)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if isinstance(a, (list, tuple)):
        lo, hi = shard_range(len(a), rank, world)
        a, b = a[lo:hi], b[lo:hi]
    if local_batch is not None:
        local = torch.from_numpy(np.ascontiguousarray(local_batch(a, b, k, costs), dtype=np.int32))
    else:
        from . import batch as B
        sa = a if isinstance(a, B.Strings) else B.Strings.from_list(a)
        sb = b if isinstance(b, B.Strings) else B.Strings.from_list(b)
        local = B.levenshtein_k_batch(sa, sb, k, costs) if sa.n else torch.empty(0, dtype=torch.int32, device=sa.blob.device)
    return all_gather_results(local, group) if gather else local


def _pairs_sharded(a, b, gather, group, run_local, run_hip):
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if isinstance(a, (list, tuple)):
        lo, hi = shard_range(len(a), rank, world)
        a, b = a[lo:hi], b[lo:hi]
    if run_local is not None:
        local = torch.from_numpy(np.ascontiguousarray(run_local(a, b), dtype=np.int32))
    else:
        from . import batch as B
        sa = a if isinstance(a, B.Strings) else B.Strings.from_list(a)
        sb = b if isinstance(b, B.Strings) else B.Strings.from_list(b)
        local = run_hip(B, sa, sb) if sa.n else torch.empty(0, dtype=torch.int32, device=sa.blob.device)
    return all_gather_results(local, group) if gather else local


def levenshtein_exp_batch_sharded(a, b, costs=LEVENSHTEIN_COSTS, gather=True, group=None, local_batch=None):
    """levenshtein_exp_with_opts(a_i, b_i, false, costs) over a pair batch sharded as levenshtein_k_batch_sharded (contiguous pairs per rank,
    no data-path collective; src/levenshtein.rs:1480-1494).  `local_batch(a_list, b_list, costs) -> int32 array` replaces the HIP path."""
    costs = _costs(costs)
    return _pairs_sharded(a, b, gather, group, None if local_batch is None else (lambda x, y: local_batch(x, y, costs)),
                          lambda B, sa, sb: B.levenshtein_exp_batch(sa, sb, costs))


def hamming_batch_sharded(a, b, gather=True, group=None, local_batch=None):
    """hamming(a_i, b_i) over a pair batch sharded the same way (src/hamming.rs:390; -1 where the lengths differ -- the reference panics)."""
    return _pairs_sharded(a, b, gather, group, local_batch, lambda B, sa, sb: B.hamming_batch(sa, sb))


def levenshtein_trace_batch_sharded(a, b, k, costs=LEVENSHTEIN_COSTS, gather=True, group=None, local_batch=None):
    """levenshtein_simd_k_with_opts(.., trace_on = true, ..) over a pair batch sharded as levenshtein_k_batch_sharded: -> (distances,
    edits (n, 2 k + 1, 2) int64, n_edits) -- every rank's records have the same `cap` = 2 k + 1 runs per pair (a script of cost <= k has
    no more), so the gathered tensors are plain concatenations in pair order.  `local_batch(a_list, b_list, k, costs, cap)` ->
    (int32 distances, int64 (n, cap, 2) edits, int32 n_edits) replaces the HIP path (tests, gloo)."""
    costs = _costs(costs)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    cap = 2 * int(k) + 1
    if isinstance(a, (list, tuple)):
        lo, hi = shard_range(len(a), rank, world)
        a, b = a[lo:hi], b[lo:hi]
    if local_batch is not None:
        d, e, ne = local_batch(a, b, k, costs, cap)
        d, e, ne = torch.from_numpy(np.ascontiguousarray(d, dtype=np.int32)), torch.from_numpy(np.ascontiguousarray(e, dtype=np.int64)), torch.from_numpy(np.ascontiguousarray(ne, dtype=np.int32))
    else:
        from . import batch as B
        sa = a if isinstance(a, B.Strings) else B.Strings.from_list(a)
        sb = b if isinstance(b, B.Strings) else B.Strings.from_list(b)
        if sa.n:
            d, e, ne = B.levenshtein_trace_batch(sa, sb, k, costs, cap=cap)
        else:
            dev = sa.blob.device
            d, e, ne = torch.empty(0, dtype=torch.int32, device=dev), torch.empty((0, cap, 2), dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.int32, device=dev)
    if not gather:
        return d, e, ne
    return all_gather_results(d, group), all_gather_results(e.reshape(-1), group).reshape(-1, cap, 2), all_gather_results(ne, group)


_MATCH_DT = np.dtype([("start", "<u8"), ("end", "<u8"), ("k", "<u4"), ("pad_", "<u4")])


def fold_best(hits, k, overlap_fold=True):
    """Sequential Best pass (src/levenshtein.rs:1792-1835) over (start, end, k) rows sorted by end
    (list of tuples or an (n, 3) integer array)."""
    rows = np.asarray(hits, dtype=np.int64).reshape(-1, 3)
    arr = np.zeros(max(len(rows), 1), dtype=_MATCH_DT)
    arr["start"][: len(rows)] = rows[:, 0]
    arr["end"][: len(rows)] = rows[:, 1]
    arr["k"][: len(rows)] = rows[:, 2]
    m = _n.lib().ta_search_fold_best(arr.ctypes.data_as(_C.POINTER(_n.MatchC)), len(rows), k, int(overlap_fold))
    return [(int(r["start"]), int(r["end"]), int(r["k"])) for r in arr[:m]]


def _gpu_local_search(needle, hay, k, costs, base, emit_from, best=False):
    """hay: (CUDA uint8 tensor with read slack, length) -- already resident, nothing is uploaded here."""
    from . import batch as B
    if best:                     # only the hits with the shard's smallest k leave the device (see below)
        return B.levenshtein_search_best_dev(needle, hay, k, costs, base, emit_from)
    return B.levenshtein_search_dev(needle, hay, k, costs, False, base, emit_from)


def _as_device_shard(shard):
    """(tensor, length) for a device-resident shard (a uint8 CUDA tensor with >= 16 bytes of read slack after `length`, or
    that pair itself); None for host bytes."""
    if isinstance(shard, tuple) and len(shard) == 2 and isinstance(shard[0], torch.Tensor):
        return shard[0], int(shard[1])
    if isinstance(shard, torch.Tensor):
        from .batch import SLACK
        return shard, shard.numel() - SLACK
    return None


def levenshtein_search_sharded(needle, shard, k, search_type=SearchType.Best, costs=LEVENSHTEIN_COSTS, group=None,
                               local_search=None):
    """levenshtein_search_simd_with_opts (unanchored) over the concatenation of every rank's `shard`.

    `shard` is host bytes, or -- the form the 1 GiB-per-GPU configuration uses -- a haystack already RESIDENT in HBM:
    a uint8 CUDA tensor with >= 16 bytes of read slack, or `(tensor, length)` (batch.haystack_tensor).  A resident shard
    is never copied or re-uploaded: the only bytes that move are the `needle_len + unit_k + 2`-byte tails every rank
    publishes (one all-gather, on the device under RCCL) and the final match lists.  Each rank runs two searches:
      * the HEAD: [left context from the previous rank(s) | the first `halo` bytes of its shard] -- a buffer of at most
        2 * halo bytes assembled on the device -- which owns the end positions inside those first `halo` bytes;
      * the BODY: the shard itself, in place, which owns every later end position (its own first `halo` bytes are the
        left context those need; a DP started fresh `halo` bytes early is exact for every cost <= k).
    Returns the same list of Match on every rank.  `local_search(needle, bytes, k, costs, base, emit_from)` ->
    int64 rows (start, end, k) replaces the HIP kernels (tests inject a CPU stand-in to exercise the partition / halo /
    gather logic under gloo; host-bytes shards only)."""
    needle = bytes(needle)
    costs = _costs(costs)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = _device_for_group(group)
    if len(needle) == 0:
        return []
    dshard = _as_device_shard(shard)
    if dshard is None:
        shard = bytes(shard)
        if local_search is None:                          # host bytes + HIP kernels: upload once, then the resident path
            from . import batch as B
            dshard = B.haystack_tensor(shard)
    elif local_search is not None:
        raise ValueError("local_search stand-ins take host bytes")
    slen = dshard[1] if dshard is not None else len(shard)
    unit_k = max(0, k - costs.start_gap_cost) // costs.gap_cost
    halo = len(needle) + unit_k + 2

    # shard lengths -> global offsets
    ln = torch.tensor([slen], dtype=torch.int64, device=dev)
    lens = [torch.zeros_like(ln) for _ in range(world)]
    dist.all_gather(lens, ln, group=group)
    lens = [int(x.item()) for x in lens]
    offs = np.concatenate([[0], np.cumsum(lens)])
    # every rank publishes its last `halo` bytes (right-aligned); a rank's left context is the tail of what precedes it
    tl = min(halo, slen)
    if dshard is not None:
        tail_t = torch.zeros(halo, dtype=torch.uint8, device=dshard[0].device)
        if tl:
            tail_t[halo - tl:] = dshard[0][slen - tl:slen]
        tail_t = tail_t.to(dev)                           # stays on the device under RCCL; `halo` bytes to the host under gloo
    else:
        tail = np.zeros(halo, dtype=np.uint8)
        if tl:
            tail[halo - tl:] = np.frombuffer(shard[-tl:], dtype=np.uint8)
        tail_t = torch.from_numpy(tail).to(dev)
    tails = [torch.zeros_like(tail_t) for _ in range(world)]
    dist.all_gather(tails, tail_t, group=group)
    pieces, have, r = [], 0, rank - 1
    while r >= 0 and have < halo:
        tr = min(halo, lens[r])
        if tr:
            pieces.insert(0, tails[r][halo - tr:])
            have += tr
        r -= 1
    ctx_t = torch.cat(pieces)[-halo:] if pieces else torch.zeros(0, dtype=torch.uint8, device=dev)
    nctx = int(ctx_t.numel())

    off = int(offs[rank])
    h1 = min(halo, slen)                                  # end positions (off, off + h1] belong to the head search
    best = search_type == SearchType.Best
    if dshard is not None:
        from .batch import SLACK
        sdev = dshard[0].device
        head = torch.zeros(nctx + h1 + SLACK, dtype=torch.uint8, device=sdev)
        if nctx:
            head[:nctx] = ctx_t.to(sdev)
        if h1:
            head[nctx:nctx + h1] = dshard[0][:h1]
        parts = []
        if h1:
            parts.append(_gpu_local_search(needle, (head, nctx + h1), k, costs, off - nctx, off, best))
        if slen > h1:
            parts.append(_gpu_local_search(needle, dshard, k, costs, off, off + h1, best))
    else:
        ctx = ctx_t.cpu().numpy().tobytes()
        parts = []
        if h1:
            parts.append(local_search(needle, ctx + shard[:h1], k, costs, off - nctx, off))
        if slen > h1:
            parts.append(local_search(needle, shard, k, costs, off, off + h1))
    parts = [np.asarray(x, dtype=np.int64).reshape(-1, 3) for x in parts]
    local = np.concatenate(parts) if parts else np.empty((0, 3), dtype=np.int64)
    if best and len(local):
        # Best keeps the hits with the globally smallest k, and once the running minimum has reached it no other hit is
        # emitted any more (src/levenshtein.rs:1792-1835): a shard only needs to contribute the hits with ITS smallest k
        local = local[local[:, 2] == local[:, 2].min()]

    # the one exchange step: gather the (tiny) match lists
    flat = torch.from_numpy(np.ascontiguousarray(local).reshape(-1)).to(dev)
    allhits = all_gather_results(flat, group).cpu().numpy().reshape(-1, 3)
    hits = [tuple(int(v) for v in row) for row in allhits]
    whole_gap = len(needle) * costs.gap_cost + costs.start_gap_cost       # the end == 0 match (:1693-1706)
    if whole_gap <= k:
        hits.insert(0, (0, 0, whole_gap))
    if search_type == SearchType.Best:
        hits = fold_best(hits, k, True)
    return [Match(*h) for h in hits]


def hamming_search_sharded(needle, shard, k, search_type=SearchType.Best, group=None, local_search=None):
    """hamming_search_simd_with_opts (src/hamming.rs:454-554) over the concatenation of every rank's `shard`.

    A window belongs to the rank its START lies in and needs `needle_len - 1` bytes of the ranks behind it: every rank publishes its first
    `needle_len - 1` bytes (one all-gather of that many bytes, on the device under RCCL).  Each rank then runs
      * the BODY: its shard in place (host bytes, or a haystack RESIDENT in HBM -- never copied or re-uploaded), the windows that lie inside it;
      * the TAIL: [its last needle_len - 1 bytes | the following ranks' first bytes], at most 2 (needle_len - 1) bytes assembled on the
        device, the windows that start in those last bytes.
    The SIMD contract's NUL rule (:463: a zero byte ANYWHERE in the haystack panics) is agreed between the ranks before anyone raises
    (one all-reduce of a flag), so no rank is left waiting in a collective.  The one exchange step of the data path is the gather of the
    match lists; the Best pass (running minimum, no overlap fold, :122-143) runs identically on every rank.
    `local_search(needle, bytes, k, base)` -> int64 rows (start, end, k) of ALL windows with <= k mismatches replaces the HIP kernels
    (tests, gloo; host-bytes shards only)."""
    from . import PanicError
    needle = bytes(needle)
    n = len(needle)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = _device_for_group(group)
    dshard = _as_device_shard(shard)
    if dshard is None:
        shard = bytes(shard)
        if local_search is None:
            from . import batch as B
            dshard = B.haystack_tensor(shard)
    elif local_search is not None:
        raise ValueError("local_search stand-ins take host bytes")
    slen = dshard[1] if dshard is not None else len(shard)
    ln = torch.tensor([slen], dtype=torch.int64, device=dev)
    lens = [torch.zeros_like(ln) for _ in range(world)]
    dist.all_gather(lens, ln, group=group)
    lens = [int(x.item()) for x in lens]
    total = sum(lens)
    if n == 0 or n > total:                               # :455-461
        return []
    offs = np.concatenate([[0], np.cumsum(lens)])
    ov = n - 1
    # every rank publishes its first `ov` bytes (left-aligned); a rank's right context is the head of what follows it
    hl = min(ov, slen)
    head_t = torch.zeros(max(ov, 1), dtype=torch.uint8, device=dshard[0].device if dshard is not None else "cpu")
    if hl:
        head_t[:hl] = dshard[0][:hl] if dshard is not None else torch.from_numpy(np.frombuffer(shard[:hl], dtype=np.uint8).copy())
    head_t = head_t.to(dev)
    heads = [torch.zeros_like(head_t) for _ in range(world)]
    dist.all_gather(heads, head_t, group=group)
    pieces, have, r = [], 0, rank + 1
    while r < world and have < ov:
        take = min(ov - have, min(ov, lens[r]))
        if take:
            pieces.append(heads[r][:take])
            have += take
        r += 1
    ctx_t = torch.cat(pieces) if pieces else torch.zeros(0, dtype=torch.uint8, device=dev)
    off = int(offs[rank])
    tl = min(ov, slen)                                    # windows that start in the shard's last `tl` bytes reach into the context
    nul, parts = 0, []
    if dshard is not None:
        from . import batch as B
        from .batch import SLACK
        sdev = dshard[0].device
        try:
            if slen >= n:
                parts.append(B.hamming_search_dev(needle, dshard, k, base=off))
            elif slen and bool((dshard[0][:slen] == 0).any().item()):
                nul = 1                                   # (a shard too short to hold a window is still part of the haystack: :463)
            if tl and tl + have >= n:
                tail = torch.zeros(tl + have + SLACK, dtype=torch.uint8, device=sdev)
                tail[:tl] = dshard[0][slen - tl:slen]
                if have:
                    tail[tl:tl + have] = ctx_t.to(sdev)
                rows = B.hamming_search_dev(needle, (tail, tl + have), k, base=off + slen - tl)
                parts.append(rows[rows[:, 0] >= off + max(slen - n + 1, 0)] if len(rows) else rows)   # (the body owns the windows inside the shard)
        except PanicError:
            nul = 1
    else:
        if b"\0" in shard:
            nul = 1
        ext = shard + ctx_t.cpu().numpy().tobytes()
        if len(ext) >= n and slen:
            rows = np.asarray(local_search(needle, ext, k, off), dtype=np.int64).reshape(-1, 3)
            parts.append(rows[rows[:, 0] < off + slen] if len(rows) else rows)
    flag = torch.tensor([nul], dtype=torch.int64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()):
        raise PanicError("No zero/null bytes allowed in the string!")
    parts = [np.asarray(x, dtype=np.int64).reshape(-1, 3) for x in parts]
    local = np.concatenate(parts) if parts else np.empty((0, 3), dtype=np.int64)
    if len(local):
        local = local[np.argsort(local[:, 1], kind="stable")]
        if search_type == SearchType.Best:                # only the hits with the shard's smallest k can survive the fold
            local = local[local[:, 2] == local[:, 2].min()]
    flat = torch.from_numpy(np.ascontiguousarray(local).reshape(-1)).to(dev)
    allhits = all_gather_results(flat, group).cpu().numpy().reshape(-1, 3)
    hits = [tuple(int(v) for v in row) for row in allhits]
    if search_type == SearchType.Best:
        hits = fold_best(hits, k, False)
    return [Match(*h) for h in hits]
