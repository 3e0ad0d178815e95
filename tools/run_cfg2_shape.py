"""BASELINE configs[1]'s shape on one GPU: a ~70 k-triangle mesh (the displaced sphere standing in for the bunny),
1024^2 atlas, 64 spp, cosine sampler, any-hit, float sums — where the stages around the trace kernel weigh most."""
import sys, time; sys.path.insert(0, '.')
from aobaker_amd import BakeEngine, Baker, capi, scenes
from aobaker_amd.scene import FlatScene
fs = FlatScene(scenes.sphere_scene(n=187, atlas=1024, seed=2, bpc=4))
BakeEngine.Init(0)
o = capi.default_options(); o.sample_count = 64; o.sampler = capi.SAMPLER_COSINE; o.accumulate = capi.ACCUMULATE_FLOAT; o.normal_fp16 = capi.FP16_OFF; o.seed = 2
b = Baker(fs, o)
for it in range(4):
    t = time.time(); b.build_bvh(); t1 = time.time(); b.rasterize(0); b.sync(); t2 = time.time(); b.trace(0); b.resolve(0); b.sync(); dt = time.time() - t
    st = b.stats()
    print(f"cfg2 shape: step {dt*1e3:.2f} ms (bvh {1e3*(t1-t):.2f} incl. host, raster {1e3*(t2-t1):.2f}, trace {b.last_trace_ms():.2f}) rays {st.ray_count} -> {st.ray_count/dt/1e9:.2f} Grays/s", flush=True)
