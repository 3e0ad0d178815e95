"""Multi-GPU sharding of the bake: one process per GPU, texel strips gathered to rank 0.

The path shards naturally: surfels are independent, the scene + BVH are
replicated on every GPU (built redundantly — cheaper than a broadcast at these sizes), the
surfel list is split into `world` contiguous ranges, and the only exchange step is ONE gather of
the per-texel {total, weight} cells to rank 0, which then resolves and bleeds.  Because a
section's surfel list is row-major, rank r's cells form a contiguous strip of the result buffer
and strips of different ranks are disjoint: rank 0 receives each peer's strip straight into its
own buffer (point-to-point over that peer's own xGMI link; no ring, nobody else needs the image).
Materials with several overlapping sections fall back to a sum-reduce.

`torch.distributed` carries the transfers: backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU
tests.  Nothing here touches torch unless it is called.
"""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced split of [0,total) — equal ray counts per rank to first order."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def weak_scaling_atlas(base: int, world: int) -> tuple[int, int]:
    """(width, height) with width*height = world * base^2, as square as possible:
    1 -> 1x1, 2 -> 1x2, 4 -> 2x2 (BASELINE cfg4's 4096^2 for base 2048), 8 -> 2x4."""
    a = 1
    while (a * 2) * (a * 2) <= world:
        a *= 2
    b = world // a
    if a * b != world:
        a, b = 1, world
    return base * a, base * b


def equal_rows(height: int, parts: int, align: int = 8) -> list[int]:
    """Cuts of `height` texel rows into `parts` strips of (nearly) equal numbers of 8-texel tile rows: parts + 1 row
    numbers, [0, ..., height].  (aob_equal_rows of include/aobaker_multi.h computes the same.)"""
    tile_rows = (height + align - 1) // align
    b = [min(tile_rows * r // parts * align, height) for r in range(parts + 1)]
    b[parts] = height
    return b


def rebalance_rows(bounds: list[int], old_ms: list[float], height: int, align: int = 8) -> list[int]:
    """New tile-row aligned cuts such that every strip costs the same, with the cost per row of old strip r taken as
    old_ms[r] / rows(r) (piecewise linear cumulative cost).  The border strips of the terrain cost less per row than the
    middle ones; two or three warm-up steps of measure -> re-cut level the ranks.  The result cells do not depend on
    where the cuts are.  Twin of aob_rebalance_rows (aobaker_amd/host/MultiGpu.cpp): same arithmetic, same cuts."""
    parts = len(old_ms)
    if len(bounds) != parts + 1 or parts == 0:
        return list(bounds)
    cost = [max(0.0, float(t)) for t in old_ms]
    total = sum(cost)
    if not total > 0.0:
        return list(bounds)
    out = [0] * (parts + 1)
    out[parts] = height
    seg, before = 0, 0.0
    for r in range(1, parts):
        want = total * r / parts
        while seg + 1 < parts and before + cost[seg] < want:
            before += cost[seg]
            seg += 1
        span = float(bounds[seg + 1] - bounds[seg])
        row = bounds[seg] + span * (want - before) / cost[seg] if cost[seg] > 0.0 else float(bounds[seg + 1])
        cut = int(row / align + 0.5) * align
        out[r] = min(max(cut, out[r - 1]), height)
    return out


def row_window(height: int, rank: int, world: int, align: int = 8) -> tuple[int, int]:
    """Rank `rank`'s strip of texel rows [begin, end): the atlas's tile rows (8 texels) dealt out evenly.  With
    `Baker.rasterize(rows=...)` a rank rasterises and traces its own strip only; its result cells are the contiguous
    range [begin * width, end * width).  (`Baker.shard_bounds` balances by surfel count instead, at the price of every
    rank rasterising the whole atlas.)"""
    tile_rows = (height + align - 1) // align
    b, e = shard_range(tile_rows, rank, world)
    return min(b * align, height), min(e * align, height)


def gather_strips(cells, my_range, rank: int, world: int, dst: int = 0, overlapping: bool = False, ranges=None):
    """Gather every rank's strip of `cells` (an [n, 2] int32 tensor aliasing the AobResult buffer)
    into rank `dst`'s buffer.  my_range = (first_cell, last_cell), first > last when empty.
    ranges: every rank's (first_cell, last_cell) when the caller knows them (row windows are a pure
    function of the rank) — saves the all-gather of the ranges and its host synchronisation."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return
    if overlapping:
        # several sections may write the same cell from different ranks: integer/float add of
        # disjoint-or-zero contributions; x + 0 is exact either way
        dist.reduce(cells, dst=dst, op=dist.ReduceOp.SUM)
        return
    if ranges is None:
        mine = torch.tensor([my_range[0], my_range[1]], dtype=torch.int64, device=cells.device)
        ranges = [torch.zeros(2, dtype=torch.int64, device=cells.device) for _ in range(world)]
        dist.all_gather(ranges, mine)
        ranges = [tuple(int(v) for v in r.tolist()) for r in ranges]
    ops = []
    if rank == dst:
        for peer, (c0, c1) in enumerate(ranges):
            if peer != dst and c0 <= c1:
                ops.append(dist.P2POp(dist.irecv, cells[c0:c1 + 1], peer))
    else:
        c0, c1 = ranges[rank]
        if c0 <= c1:
            ops.append(dist.P2POp(dist.isend, cells[c0:c1 + 1], dst))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


class DeviceAlias:
    """Zero-copy torch view of a raw HIP allocation (the C ABI hands out plain pointers)."""

    def __init__(self, ptr: int, nbytes: int, typestr: str = "<i4"):
        itemsize = int(typestr[2:])
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr,
                                         "data": (ptr, False), "version": 2}


def alias_results(baker, material: int = 0, as_float: bool = False):
    """[cells, 2] tensor over the material's AobResult buffer: int32 cells in QUANTISED mode,
    float32 in FLOAT mode (only the sum-reduce fallback cares about the type)."""
    import torch
    ptr, nbytes = baker.results_device_ptr(material)
    t = torch.as_tensor(DeviceAlias(ptr, nbytes, "<f4" if as_float else "<i4"), device="cuda")
    return t.view(-1, 2)
