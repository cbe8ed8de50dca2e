"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).

The hot path shards without any data-path collective:
  * pair batches: contiguous N/G pairs per rank; results stay on the rank (or are all-gathered on request);
  * levenshtein_search over one big haystack: contiguous shards with a left halo of needle_len + unit_k + 2
    bytes taken from the previous rank(s); every rank emits All-mode hits for its own end positions only; the
    ONE exchange step is the final match-list gather (counts, then padded records), after which the
    order-dependent Best fold runs identically on every rank.
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


def _gpu_local_search(needle, hay_ext, k, costs, base, emit_from, best=False):
    from . import batch as B
    t = B.haystack_tensor(hay_ext) if not isinstance(hay_ext, tuple) else hay_ext
    if best:                     # only the hits with the shard's smallest k leave the device (see below)
        return B.levenshtein_search_best_dev(needle, t, k, costs, base, emit_from)
    return B.levenshtein_search_dev(needle, t, k, costs, False, base, emit_from)


def levenshtein_search_sharded(needle, shard, k, search_type=SearchType.Best, costs=LEVENSHTEIN_COSTS, group=None,
                               local_search=None):
    """levenshtein_search_simd_with_opts (unanchored) over the concatenation of every rank's `shard` (bytes).

    Returns the same list of Match on every rank.  `local_search(needle, bytes, k, costs, base, emit_from)` ->
    int64 rows (start, end, k) defaults to the HIP kernel; tests inject a CPU stand-in to exercise the
    partition / halo / gather logic under gloo."""
    needle = bytes(needle)
    shard = bytes(shard)
    costs = _costs(costs)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = _device_for_group(group)
    if len(needle) == 0:
        return []
    unit_k = max(0, k - costs.start_gap_cost) // costs.gap_cost
    halo = len(needle) + unit_k + 2

    # shard lengths -> global offsets
    ln = torch.tensor([len(shard)], dtype=torch.int64, device=dev)
    lens = [torch.zeros_like(ln) for _ in range(world)]
    dist.all_gather(lens, ln, group=group)
    lens = [int(x.item()) for x in lens]
    offs = np.concatenate([[0], np.cumsum(lens)])
    # every rank publishes its last `halo` bytes; a rank's left context is the tail of what precedes it
    tail = np.zeros(halo, dtype=np.uint8)
    tl = min(halo, len(shard))
    if tl:
        tail[halo - tl:] = np.frombuffer(shard[-tl:], dtype=np.uint8)
    tail_t = torch.from_numpy(tail).to(dev)
    tails = [torch.zeros_like(tail_t) for _ in range(world)]
    dist.all_gather(tails, tail_t, group=group)
    ctx = b""
    r = rank - 1
    while r >= 0 and len(ctx) < halo:
        tr = min(halo, lens[r])
        piece = tails[r].cpu().numpy().tobytes()[halo - tr:] if tr else b""
        ctx = piece + ctx
        r -= 1
    ctx = ctx[-halo:] if len(ctx) > halo else ctx

    base = int(offs[rank]) - len(ctx)
    best = search_type == SearchType.Best
    if local_search is None:
        local = _gpu_local_search(needle, ctx + shard, k, costs, base, int(offs[rank]), best)
    else:
        local = local_search(needle, ctx + shard, k, costs, base, int(offs[rank]))
    local = np.asarray(local, dtype=np.int64).reshape(-1, 3)
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
