"""The N>1 path on CPU: two gloo ranks shard the surfel list, trace with the oracle standing in
for the GPU stage, and gather their texel strips to rank 0 with the SAME host code
(aobaker_amd.dist.gather_strips) the RCCL path uses."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, overlapping, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aobaker_amd import capi, scenes
    from aobaker_amd import dist as aod
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc

    sc = scenes.two_section_scene(32) if overlapping else scenes.cube_scene(64)
    flat = FlatScene(sc)
    m = flat.material_array[0]
    opt = orc.default_options()
    opt.sample_count = 8
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    hemi = orc.compat_table(opt)
    begin, end = aod.shard_range(len(surfels), rank, world)
    mine, _ = orc.trace(nodes, surfels, sidx, opt, m.width, m.height, hemi, begin=begin, end=end)
    cells = torch.from_numpy(mine.view(np.int32).reshape(-1, 2).copy())
    if end > begin:
        c = surfels["texel_y"][begin:end].astype(np.int64) * m.width + surfels["texel_x"][begin:end]
        rng = (int(c[0]), int(c[-1])) if not overlapping else (int(c.min()), int(c.max()))
    else:
        rng = (1, 0)
    aod.gather_strips(cells, rng, rank, world, 0, overlapping)
    if rank == 0:
        full, _ = orc.trace(nodes, surfels, sidx, opt, m.width, m.height, hemi)
        ok = np.array_equal(cells.numpy(), full.view(np.int32).reshape(-1, 2))
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlapping", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_trace_gathers_to_the_single_rank_result(tmp_path, world, overlapping):
    port = 29500 + (os.getpid() + world * 7 + int(overlapping)) % 2000
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, port, overlapping, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


def _row_worker(rank, world, port, out_path):
    """bench.py's N>1 scheme: every rank rasterises and traces its own strip of texel rows (dist.row_window), then the
    strips are gathered to rank 0 — here with the oracle standing in for the GPU stages."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aobaker_amd import scenes
    from aobaker_amd import dist as aod
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc

    flat = FlatScene(scenes.terrain_scene(grid=24, atlas=(32, 40)))
    m = flat.material_array[0]
    opt = orc.default_options()
    opt.sample_count = 8
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    hemi = orc.compat_table(opt)
    y0, y1 = aod.row_window(m.height, rank, world)
    keep = (surfels["texel_y"] >= y0) & (surfels["texel_y"] < y1)            # what aob_bake_rasterize_rows leaves
    mine, _ = orc.trace(nodes, np.ascontiguousarray(surfels[keep]), np.ascontiguousarray(sidx[keep]), opt, m.width, m.height, hemi)
    cells = torch.from_numpy(mine.view(np.int32).reshape(-1, 2).copy())
    every = [(b * m.width, e * m.width - 1) for b, e in (aod.row_window(m.height, r, world) for r in range(world))]
    aod.gather_strips(cells, every[rank], rank, world, 0, False, every)          # as bench.py calls it: no all-gather of the ranges
    if rank == 0:
        full, _ = orc.trace(nodes, surfels, sidx, opt, m.width, m.height, hemi)
        ok = np.array_equal(cells.numpy(), full.view(np.int32).reshape(-1, 2))
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_row_window_strips_gather_to_the_single_rank_result(tmp_path, world):
    port = 31500 + (os.getpid() + world * 11) % 2000
    out = tmp_path / "result.txt"
    mp.spawn(_row_worker, args=(world, port, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"
