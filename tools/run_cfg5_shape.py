import sys, time; sys.path.insert(0, '.')
import numpy as np
from aobaker_amd import BakeEngine, Baker, capi, scenes
from aobaker_amd.scene import FlatScene
t=time.time(); sc = scenes.instanced_scene(instances=100, rock_n=224, atlas=8192, seed=5, bpc=1); fs = FlatScene(sc); print("scene", round(time.time()-t,1), "s, tris", fs.triangle_count(), flush=True)
BakeEngine.Init(0)
o = capi.default_options(); o.sample_count = 64; o.sampler = capi.SAMPLER_COSINE; o.accumulate = capi.ACCUMULATE_FLOAT; o.use_any_hit = 0; o.clamp_tmax_to_falloff = 1; o.falloff_distance = 1.25; o.seed = 5
t=time.time(); b = Baker(fs, o); print("upload", round(time.time()-t,2), flush=True)
for it in range(2):
    t=time.time(); b.build_bvh(); b.rasterize(0); b.trace(0); b.resolve(0); b.sync(); dt=time.time()-t
    st = b.stats(); print(f"step {dt*1e3:.1f} ms  bvh {st.ms_bvh_build:.2f} ms depth {st.bvh_depth} nodes {st.node_count} surfels {st.surfel_count} rays {st.ray_count} trace {b.last_trace_ms():.1f} ms -> {st.ray_count/b.last_trace_ms()/1e6:.2f} Grays/s", flush=True)
img = np.zeros(8192*8192, dtype=np.uint8); b.download(0, img); print("mean ao", img.mean()/255, "covered", (img>0).mean())
