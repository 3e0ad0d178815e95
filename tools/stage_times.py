#!/usr/bin/env python3
"""tools/stage_times.py — HIP-event times of the stages of one BASELINE configs[2] bake (aob_bake_last_stage_ms): the tree's build, the rasterisation, the
trace stage and, inside it, the tile cuts, the horizon tables and k_trace; best of four traces.  AOBAKER_HIP_LIBRARY picks an A/B build
(tools/build_variants.py).  With a number N: the whole step N times, every stage's time per step.  Product code only."""
import sys, os, json
sys.path.insert(0, os.getcwd())
from aobaker_amd import Baker, capi, scenes, BakeEngine
from aobaker_amd.scene import FlatScene
BakeEngine.Init(0)
flat = FlatScene(scenes.terrain_scene(grid=708, atlas=2048, seed=3, bpc=4))
o = capi.default_options(); o.sample_count=256; o.sampler=capi.SAMPLER_COSINE; o.seed=3; o.accumulate=capi.ACCUMULATE_FLOAT; o.normal_fp16=capi.FP16_OFF
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
with Baker(flat, o) as b:
    if steps:
        # the bench's own step, `steps` times in a row: does a stage's time drift as the card warms up?
        hist = []
        for _ in range(steps):
            b.build_bvh(); b.rasterize(0); b.trace(0); b.resolve(0); b.sync(); hist.append(b.last_stage_ms())
        for k in ("bvh_build", "rasterize", "tile_cuts", "horizon", "walk"):
            print(json.dumps({"stage": k, "ms": [round(h[k], 2) for h in hist if h.get(k) is not None]}), flush=True)
        sys.exit(0)
    b.build_bvh(); b.rasterize(0)
    best=None
    for _ in range(4):
        b.clear_results(0); b.trace(0); b.sync(); s=b.last_stage_ms()
        if best is None or s["trace"] < best["trace"]: best=s
    st=b.stats()
    print(json.dumps({"lib": os.environ.get("AOBAKER_HIP_LIBRARY","in-tree").split("/")[-1], **{k: round(v,3) for k,v in best.items() if v is not None}, "cast_share": round(st.rays_cast/ max(1,st.ray_count),4)}), flush=True)
