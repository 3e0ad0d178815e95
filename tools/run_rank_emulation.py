"""What ONE rank of `bench.py --gpus N` does, on one GPU: the N x 2048^2 atlas of the weak-scaling run, rank r's strip of
texel rows rasterised and traced (no gather: that needs the peers).  Shows what the per-rank step costs as N grows —
the figure DESIGN.md §7 quotes.  Usage: python tools/run_rank_emulation.py [N ...]"""
import sys, time; sys.path.insert(0, '.')
from aobaker_amd import BakeEngine, Baker, capi, scenes
from aobaker_amd import dist as aod
from aobaker_amd.scene import FlatScene
BakeEngine.Init(0)
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    W, H = aod.weak_scaling_atlas(2048, world)
    flat = FlatScene(scenes.terrain_scene(grid=708, atlas=(W, H), seed=3, bpc=4))
    o = capi.default_options(); o.sample_count = 256; o.sampler = capi.SAMPLER_COSINE; o.seed = 3
    o.use_any_hit = 1; o.accumulate = capi.ACCUMULATE_FLOAT; o.normal_fp16 = capi.FP16_OFF
    with Baker(flat, o) as b:
        line = []
        for rank in sorted({0, world // 2, world - 1}):
            rows = aod.row_window(H, rank, world)
            best = 1e9
            for it in range(3):
                t = time.time(); b.build_bvh(); b.rasterize(0, rows if world > 1 else None); n = b.surfel_count(0); b.trace(0, 0, n)
                if rank == 0: b.resolve(0)
                b.sync(); best = min(best, time.time() - t)
            line.append(f"rank {rank}: {best*1e3:.1f} ms ({n * 256 / best / 1e9:.2f} Grays/s)")
        print(f"N={world} atlas {W}x{H}: " + ", ".join(line), flush=True)
