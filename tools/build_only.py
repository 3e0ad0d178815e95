import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aobaker_amd import BakeEngine, Baker, capi, scenes
from aobaker_amd.scene import FlatScene
BakeEngine.Init(0)
flat = FlatScene(scenes.terrain_scene(grid=708, atlas=256, seed=3, bpc=4))
o = capi.default_options()
o.sampler, o.accumulate, o.normal_fp16 = capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF
o.bvh_builder = int(sys.argv[1]) if len(sys.argv) > 1 else 2
with Baker(flat, o) as b:
    for _ in range(5):
        b.build_bvh(); b.sync()
BakeEngine.Dispose()
