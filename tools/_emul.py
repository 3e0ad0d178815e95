import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aobaker_amd import BakeEngine, Baker, capi, scenes
from aobaker_amd import dist as aod
from aobaker_amd.scene import FlatScene
world = 8
W, H = aod.weak_scaling_atlas(2048, world)
flat = FlatScene(scenes.terrain_scene(grid=708, atlas=(W, H), seed=3, bpc=4))
BakeEngine.Init(0)
o = capi.default_options(); o.sample_count = 256; o.sampler = 1; o.seed = 3; o.accumulate = 1; o.normal_fp16 = 0
b = Baker(flat, o)
b.build_bvh(); b.rasterize(0)
for parts in (8, 32):
    bounds, strips, ov = b.shard_bounds(0, parts)
    ts = []
    for p in range(parts):
        b.trace(0, bounds[p], bounds[p + 1]); b.sync()
        ts.append(b.last_trace_ms())
    print(f"parts {parts}:", [round(t, 1) for t in ts], flush=True)
    if parts == 32:
        print("interleaved x4 per rank:", [round(sum(ts[r::8]), 1) for r in range(8)])
        print("contiguous x4 per rank:", [round(sum(ts[4*r:4*r+4]), 1) for r in range(8)])
b.close(); BakeEngine.Dispose()
