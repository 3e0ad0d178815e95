#!/usr/bin/env python3
"""tools/run_shapes.py — the BASELINE.json workload shapes other than bench.py's headline, on one GPU, as JSON lines
(commit the output under profiles/: DESIGN.md quotes these figures).

  cfg2        configs[1]: ~70 k-triangle mesh (the displaced sphere standing in for the Stanford bunny, which is not in the
              image; an OBJ can be baked with tools/aobake_cli --obj), 1024^2 atlas, 64 spp, cosine / any-hit / float
  reference   the reference's own constants (BakeEngine.cpp:13-18: 100 spp, MSVC-rand table sampler, quantised int32 sums,
              any-hit, R8 output) on the 1 M-triangle terrain at 2048^2, plus the one-shot aob_bake wall-clock in that mode
  cfg5        configs[4]: 100 x 100 352-triangle rocks + ground = 10 035 202 triangles, 8192^2 atlas, 1024 spp, closest-hit
              falloff scoring with the ray length clamped to the falloff distance 1.25 — one GPU's bake of the whole atlas
  rank-N      one rank's strip of configs[3]'s fixed 4096^2 atlas cut N ways (first, middle, last strip), what
              `bench.py --gpus N` gives every rank before the strips are re-cut by cost

  python tools/run_shapes.py [cfg2] [reference] [cfg5] [ranks] > gpurun_out/shapes.jsonl       (default: all)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def emit(**kw):
    print(json.dumps(kw), flush=True)


def timed_steps(b, n, rows=None, resolve=True):
    best = None
    for _ in range(n):
        t = time.perf_counter()
        b.build_bvh()
        t1 = time.perf_counter()
        b.rasterize(0, rows)
        b.sync()
        t2 = time.perf_counter()
        b.trace(0)
        if resolve:
            b.resolve(0)
        b.sync()
        dt = time.perf_counter() - t
        rec = {"step_ms": round(dt * 1e3, 3), "bvh_build_host_ms": round((t1 - t) * 1e3, 3), "rasterize_ms": round((t2 - t1) * 1e3, 3),
               "trace_ms": round(b.last_trace_ms(), 3)}
        if best is None or rec["step_ms"] < best["step_ms"]:
            best = rec
    return best


def main():
    which = set(sys.argv[1:]) or {"cfg2", "reference", "cfg5", "ranks"}
    import torch  # noqa: F401  (one HIP runtime)
    from aobaker_amd import BakeEngine, Baker, capi, scenes
    from aobaker_amd import dist as aod
    from aobaker_amd.scene import FlatScene
    BakeEngine.Init(0)

    def cosine(spp, seed, **kw):
        o = capi.default_options()
        o.sample_count, o.sampler, o.accumulate, o.normal_fp16, o.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, seed
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    if "cfg2" in which:
        sc2, mesh_name = scenes.config2_scene(atlas=1024, seed=2, bpc=4)        # AOB_CFG2_OBJ=/path/to.obj: that mesh instead of the stand-in
        flat = FlatScene(sc2)
        with Baker(flat, cosine(64, 2)) as b:
            rec = timed_steps(b, 5)
            st = b.stats()
        emit(shape="cfg2", workload=f"configs[1]: {st.triangle_count}-triangle mesh, 1024^2, 64 spp, cosine, any-hit, float", mesh=mesh_name,
             triangles=st.triangle_count, rays=st.ray_count, grays_per_s=round(st.ray_count / rec["step_ms"] / 1e6, 3), **rec)
    if "reference" in which:
        sc = scenes.terrain_scene(grid=708, atlas=2048, seed=3, bpc=1)
        o = capi.default_options()
        with Baker(FlatScene(sc), o) as b:
            rec = timed_steps(b, 4)
            st = b.stats()
        t = time.perf_counter()
        BakeEngine.Options = o
        BakeEngine.Bake(sc)
        one_shot = (time.perf_counter() - t) * 1e3
        emit(shape="reference", workload="the reference's constants (100 spp, MSVC-rand table, int32 quantised sums, any-hit, R8) on the 999 698-triangle terrain, 2048^2",
             rays=st.ray_count, grays_per_s=round(st.ray_count / rec["step_ms"] / 1e6, 3), aob_bake_one_shot_ms=round(one_shot, 2),
             one_shot_stats=BakeEngine.LastStats.as_dict(), **rec)
        BakeEngine.Options = None
    if "ranks" in which:
        flat = FlatScene(scenes.terrain_scene(grid=708, atlas=4096, seed=3, bpc=4))
        with Baker(flat, cosine(256, 3, bvh_builder=2)) as b:          # as bench.py bakes it
            for world in (1, 2, 4, 8):
                bounds = aod.equal_rows(4096, world)
                for rank in sorted({0, world // 2, world - 1}):
                    rows = (bounds[rank], bounds[rank + 1]) if world > 1 else None
                    rec = timed_steps(b, 3, rows, resolve=rank == 0)
                    n = b.surfel_count(0)
                    emit(shape=f"rank-{world}", rank=rank, rows=rows, surfels=n, grays_per_s=round(n * 256 / rec["step_ms"] / 1e6, 3), **rec)
    if "cfg5" in which:
        t = time.perf_counter()
        flat = FlatScene(scenes.instanced_scene(instances=100, rock_n=224, atlas=8192, seed=5, bpc=1))
        gen = time.perf_counter() - t
        with Baker(flat, cosine(1024, 5, use_any_hit=0, falloff_distance=1.25, clamp_tmax_to_falloff=1)) as b:
            rec = timed_steps(b, 2)
            st = b.stats()
        emit(shape="cfg5", workload="configs[4]: 100 x 100 352-triangle rocks + ground (every copy a full shape: flat tree), 8192^2, 1024 spp, "
                                    "closest-hit falloff scoring, tmax clamped to the falloff distance 1.25",
             triangles=st.triangle_count, bvh_nodes=st.node_count, bvh_depth=st.bvh_depth, bvh_build_device_ms=round(st.ms_bvh_build, 3),
             surfels=st.surfel_count, rays=st.ray_count, grays_per_s=round(st.ray_count / rec["step_ms"] / 1e6, 3), scene_generation_s=round(gen, 1), **rec)
    BakeEngine.Dispose()


if __name__ == "__main__":
    main()
