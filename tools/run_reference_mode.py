import sys, time; sys.path.insert(0, '.')
import numpy as np
from aobaker_amd import BakeEngine, Baker, capi, scenes
from aobaker_amd.scene import FlatScene
fs = FlatScene(scenes.terrain_scene(grid=708, atlas=2048, seed=3, bpc=1))
BakeEngine.Init(0)
o = capi.default_options()          # the reference's constants: 100 spp, MSVC-rand table sampler, quantised sums, any-hit, R8
b = Baker(fs, o)
for it in range(3):
    t=time.time(); b.build_bvh(); b.rasterize(0); b.trace(0); b.resolve(0); b.sync(); dt=time.time()-t
    st=b.stats(); print(f"reference mode: step {dt*1e3:.1f} ms trace {b.last_trace_ms():.1f} ms rays {st.ray_count} -> {st.ray_count/dt/1e9:.2f} Grays/s", flush=True)
t=time.time(); sc2 = scenes.terrain_scene(grid=708, atlas=2048, seed=3, bpc=1); BakeEngine.Options=o; BakeEngine.Bake(sc2); print("aob_bake end to end incl. upload/download/table: %.1f ms" % ((time.time()-t)*1e3)); print(BakeEngine.LastStats.as_dict())
