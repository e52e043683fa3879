"""Interval sharding of the readcount path across ranks (one process per GPU).

The path partitions by genomic interval with no exchange step: every output line depends only on the reads that
overlap its position (plus the position before it for deletions, which the engine's lead position beg0-1 carries,
bamreadcount.cpp:269/602).  So each rank takes a contiguous slice of the work list — the site list in FILE ORDER for
-l, the region list for command-line regions, or equal sub-intervals of one long region — processes it independently,
and rank 0 concatenates the per-rank text in rank order.  The only communication is the gather of the text as
size-prefixed byte tensors (and the all-reduce of two counters for the metrics line); no collective touches pileup data.  With torch.distributed backend
"nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import numpy as np

from . import capi


BIN = 16384      # the BAI linear index's window: what a caller that only has the index can weigh intervals with


def split_region(beg0, end, parts):
    """Cut [beg0,end) into `parts` abutting sub-intervals of near-equal length (empty ones dropped)."""
    edges = [beg0 + (end - beg0) * i // parts for i in range(parts + 1)]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def bin_events(arrs, length, bin_size=BIN):
    """Estimated pileup base-events per 16-kb window of a contig ("balanced by estimated event count — BAI
    linear-index / chunk sizes or a coverage prepass"): every read adds its reference span to the windows it overlaps.
    Returns the cumulative events at every window edge (len = n_bins + 1), so any interval's weight is a difference."""
    nb = int((length + bin_size - 1) // bin_size)
    # reads that never enter a column weigh nothing (unmapped, no position); a record at or past the contig's end — legal in
    # a BAM whose header understates a length — is clipped INTO the last window (pos == length would index one window past it)
    pos_all = np.asarray(arrs["pos"]).astype(np.int64)
    live = (pos_all >= 0) & ((np.asarray(arrs["flag"]).astype(np.int64) & 4) == 0) if "flag" in arrs else (pos_all >= 0)
    pos = np.clip(pos_all, 0, max(length - 1, 0))[live]
    end = np.clip(capi.read_ends(arrs)[live], pos, length)
    if nb == 0:
        return np.zeros(1, np.float64)
    # coverage as a difference array on base resolution would be length-sized; per window: spans split at window edges
    cum = np.zeros(nb + 1, np.float64)
    b0 = pos // bin_size; b1 = np.maximum(end - 1, pos) // bin_size
    same = b0 == b1
    np.add.at(cum, b0[same] + 1, (end - pos)[same])
    if (~same).any():
        p, e, x0, x1 = pos[~same], end[~same], b0[~same], b1[~same]
        np.add.at(cum, x0 + 1, (x0 + 1) * bin_size - p)                   # first window: up to its right edge
        np.add.at(cum, x1 + 1, e - x1 * bin_size)                         # last window: from its left edge
        mid = x1 - x0 - 1                                                 # whole windows in between (reads longer than a window)
        if (mid > 0).any():
            for a, m in zip(x0[mid > 0], mid[mid > 0]):
                cum[a + 2:a + 2 + m] += bin_size
    return np.cumsum(cum)


def interval_events(cum, beg0, end, bin_size=BIN):
    """estimated events of [beg0, end) from bin_events' cumulative table (linear inside a window)"""
    def at(x):
        x = min(max(x, 0), (len(cum) - 1) * bin_size)
        b = min(int(x // bin_size), len(cum) - 2) if len(cum) > 1 else 0
        return cum[b] + (cum[b + 1] - cum[b]) * (x - b * bin_size) / bin_size if len(cum) > 1 else 0.0
    return max(at(end) - at(beg0), 0.0)


def split_region_by_events(beg0, end, parts, cum, bin_size=BIN):
    """Cut [beg0,end) into `parts` abutting sub-intervals of near-equal ESTIMATED EVENTS (uneven depth: a tumour panel, a
    contig with a collapsed repeat); falls back to equal lengths where the table holds no events."""
    total = interval_events(cum, beg0, end, bin_size)
    if total <= 0 or parts <= 1:
        return split_region(beg0, end, parts)
    edges = [beg0]
    grid = np.arange(beg0, end + 1, max(1, bin_size // 16), dtype=np.int64)
    if grid[-1] != end:
        grid = np.append(grid, end)
    acc = np.array([interval_events(cum, beg0, int(x), bin_size) for x in grid])
    for i in range(1, parts):
        edges.append(int(grid[int(np.searchsorted(acc, total * i / parts, side="left"))]))
    edges.append(end)
    edges = sorted(set(edges))
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def partition(items, world, weights=None):
    """Contiguous, order-preserving slices of a work list (the -l list in file order, the command line's regions), one per
    rank, balanced by `weights` (one per item: estimated events, interval_events) — by interval length when none are given."""
    if not items:
        return [[] for _ in range(world)]
    w = np.array([max(1, e - b) for (b, e) in items], dtype=np.float64) if weights is None else np.maximum(np.asarray(weights, np.float64), 1e-9)
    assert len(w) == len(items)
    cum = np.cumsum(w)
    total = cum[-1]
    out, start = [], 0
    for r in range(world):
        # the item that crosses the r-th boundary goes to whichever side leaves the smaller excess
        if r + 1 < world:
            target = total * (r + 1) / world
            stop = int(np.searchsorted(cum, target, side="right"))
            if stop < len(items) and stop >= start and (cum[stop] - target) < (target - (cum[stop - 1] if stop > 0 else 0.0)):
                stop += 1
        else:
            stop = len(items)
        stop = min(max(stop, start), len(items))
        out.append(items[start:stop])
        start = stop
    return out


def run_sharded(lib, arrs, regions, tid, chrom, ref, dist=None, clear_queue=True, weights=None, per_rank=None, **engine_opts):
    """Every rank runs its slice of `regions` through its own engine; returns (text on rank 0 / None elsewhere,
    (events, positions) summed over ranks).  `dist` = an initialised torch.distributed module or None (single process).
    weights: one estimated event count per region (partition); per_rank: a list that receives every rank's own event count
    (gathered: the balance of the partition)."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    mine = partition(list(regions), world, weights)[rank]
    eng = capi.Engine(lib, **engine_opts)
    try:
        text, results = capi.run_regions(eng, arrs, mine, tid, chrom, ref, clear_queue=clear_queue)
    finally:
        eng.close()
    ev = sum(r.n_events for r in results)
    npos = text.count(b"\n")
    if dist is None:
        if per_rank is not None:
            per_rank[:] = [ev]
        return text, (ev, npos)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    cnt = torch.tensor([ev, npos], dtype=torch.int64, device=dev)
    if per_rank is not None:
        each = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(each, cnt)
        per_rank[:] = [int(x[0]) for x in each]
    dist.all_reduce(cnt)                                    # metrics only: 16 bytes
    return gather_text(text, dist), (int(cnt[0]), int(cnt[1]))


def gather_text(text, dist):
    """Ordered concatenation of the ranks' text on rank 0 as raw bytes: one all_gather of the sizes (8 bytes per rank), then
    one padded uint8 all_gather (text of dense regions runs to gigabytes: no pickling, no per-object round trips)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    size = torch.tensor([len(text)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(x.item()) for x in sizes]
    cap = max(max(sizes), 1)
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if len(text):
        mine[:len(text)] = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    if rank == 0:
        bufs = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(mine, bufs, dst=0)
        return b"".join(bytes(b[:n].cpu().numpy()) for b, n in zip(bufs, sizes))
    dist.gather(mine, None, dst=0)
    return None
