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


def split_region(beg0, end, parts):
    """Cut [beg0,end) into `parts` abutting sub-intervals of near-equal length (empty ones dropped)."""
    edges = [beg0 + (end - beg0) * i // parts for i in range(parts + 1)]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def partition(items, world):
    """Contiguous, order-preserving slices of a work list, balanced by summed interval length."""
    if not items:
        return [[] for _ in range(world)]
    w = np.array([max(1, e - b) for (b, e) in items], dtype=np.float64)
    cum = np.cumsum(w)
    total = cum[-1]
    out, start = [], 0
    for r in range(world):
        stop = int(np.searchsorted(cum, total * (r + 1) / world, side="right")) if r + 1 < world else len(items)
        stop = max(stop, start)
        out.append(items[start:stop])
        start = stop
    return out


def run_sharded(lib, arrs, regions, tid, chrom, ref, dist=None, clear_queue=True, **engine_opts):
    """Every rank runs its slice of `regions` through its own engine; returns (text on rank 0 / None elsewhere,
    (events, positions) summed over ranks).  `dist` = an initialised torch.distributed module or None (single process)."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    mine = partition(list(regions), world)[rank]
    eng = capi.Engine(lib, **engine_opts)
    try:
        text, results = capi.run_regions(eng, arrs, mine, tid, chrom, ref, clear_queue=clear_queue)
    finally:
        eng.close()
    ev = sum(r.n_events for r in results)
    npos = text.count(b"\n")
    if dist is None:
        return text, (ev, npos)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    cnt = torch.tensor([ev, npos], dtype=torch.int64, device=dev)
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
