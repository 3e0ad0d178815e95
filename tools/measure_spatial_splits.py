#!/usr/bin/env python3
"""tools/measure_spatial_splits.py (a measurement script: the oracle as a counter; not a test, not product code) — what a full
top-down SAH tree and SPATIAL SPLITS (the reference stack's `bvh.builder = "sah"`, `bvh.sah.use_splits`,
RadeonRays/radeon_rays.h:286-294) would buy, counted by the CPU oracle on the rays of an any-hit bake.

Trees (all BVH2 in the reference's fp32 node layout, walked by the same counter, orc_flat_intersect): PLOC (bvh_builder 1), PLOC
under a binned-SAH top (bvh_builder 2), a full top-down binned-SAH tree, the same with spatial splits (reference budget 1.3 n and
2 n).  Scenes: BASELINE configs[2]'s terrain (near-equilateral triangles: splits cannot show), a ball of long thin triangles
through one point (`hairball`), and long thin sticks scattered over a floor (`straw`) — the shapes real meshes have and the
synthetic BASELINE scenes lack.

  python tools/measure_spatial_splits.py [--scenes terrain,hairball,straw] [--n 100000] [--out FILE]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def thin_scene(kind, n, seed=11):
    """A baked floor patch (32^2 atlas) under n long thin triangles: `hairball` — all through a small ball, length 0.9, width 0.01;
    `straw` — scattered in a slab above the floor, length 0.05 .. 0.4, width 0.004."""
    from aobaker_amd import scenes
    from aobaker_amd.scene import BakeObject, RenderMesh, Section, make_vertices
    rng = np.random.default_rng(seed)
    sc = scenes.plane_scene(32, bpc=4)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    w = rng.normal(size=(n, 3)); w -= (w * d).sum(1, keepdims=True) * d; w /= np.linalg.norm(w, axis=1, keepdims=True)
    if kind == "hairball":
        o = np.float32([0.5, 0.6, 0.5]) + rng.normal(scale=0.05, size=(n, 3))
        half, width = 0.45, 0.01
    else:
        o = np.stack([rng.uniform(0, 1, n), rng.uniform(0.05, 0.6, n), rng.uniform(0, 1, n)], 1)
        half, width = rng.uniform(0.025, 0.2, (n, 1)), 0.004
    v = np.stack([o - half * d, o + half * d, o + width * w], axis=1).astype(np.float32).reshape(-1, 3)
    nrm = np.repeat(np.cross(d, w).astype(np.float32), 3, axis=0)
    uv = np.zeros((3 * n, 2), dtype=np.float32)
    obj = sc.AddObject(BakeObject(kind))
    obj.AddSection(Section("Geometry", RenderMesh(make_vertices(v, nrm, uv), np.arange(3 * n, dtype=np.uint32)), None))
    return sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", default="terrain,hairball,straw")
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc
    lib = orc.load()
    lib.orc_bvh_build_sbvh.restype = C.c_uint64
    lib.orc_bvh_build_sbvh.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]

    def sbvh(tris, budget):
        tris = np.ascontiguousarray(tris)
        n = len(tris)
        cap = max(int(budget * n), n)
        nodes = np.zeros(8 * cap + 32, dtype=capi.node_dtype())
        depth, leaves = C.c_uint32(), C.c_uint64()
        cnt = int(lib.orc_bvh_build_sbvh(tris.ctypes.data, n, nodes.ctypes.data, len(nodes), cap if budget > 1.0 else 0, 1e-5, C.byref(depth), C.byref(leaves)))
        assert cnt > 0, "the tree outgrew its node buffer"
        return nodes[:cnt].copy(), depth.value, int(leaves.value)

    out = []
    for name in args.scenes.split(","):
        if name == "terrain":
            flat = FlatScene(scenes.terrain_scene(grid=708, atlas=2048, seed=3)); atlas, spp, seed, one_in = 2048, 256, 3, 4096
        else:
            flat = FlatScene(thin_scene(name, args.n)); atlas, spp, seed, one_in = 32, args.spp, 5, 1
        opt = orc.default_options()
        opt.sample_count, opt.sampler, opt.accumulate, opt.normal_fp16, opt.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, seed
        tris = orc.scene_triangles(flat.desc)
        surfels, _ = orc.rasterize(flat.desc, 0, opt)
        if one_in > 1:
            rng = np.random.default_rng(7)
            surfels = np.ascontiguousarray(surfels[rng.choice(len(surfels), size=len(surfels) // one_in, replace=False)])
        rays = np.ascontiguousarray(orc.generate_rays(surfels, None, opt, atlas))
        rec = {"scene": name, "triangles": len(tris), "rays": len(rays), "trees": []}
        builds = [("PLOC (bvh_builder 1)", lambda: orc.bvh_build(tris, 1, 1) + (len(tris),)),
                  ("PLOC under a binned-SAH top (bvh_builder 2)", lambda: orc.bvh_build(tris, 2, 1) + (len(tris),)),
                  ("full top-down binned SAH", lambda: sbvh(tris, 1.0)),
                  ("SAH + spatial splits, budget 1.3 n", lambda: sbvh(tris, 1.3)),
                  ("SAH + spatial splits, budget 2 n", lambda: sbvh(tris, 2.0))]
        base = None
        for label, build in builds:
            t = time.perf_counter()
            nodes, depth, leaves = build()
            dt = time.perf_counter() - t
            _, st = orc.flat_intersect(nodes, rays, any_hit=True, want_hits=False)
            r = float(st.rays)
            row = {"tree": label, "leaves": leaves, "depth": depth, "inner_visits_per_ray": round(st.inner_visits / r, 3),
                   "triangle_tests_per_ray": round(st.leaf_visits / r, 3), "hits": int(st.hits), "cpu_build_s": round(dt, 2)}
            if base is None:
                base = row
            row["inner_visits_vs_PLOC"] = round(row["inner_visits_per_ray"] / base["inner_visits_per_ray"], 3)
            assert row["hits"] == base["hits"], (label, row["hits"], base["hits"])          # any tree, the same occluded rays
            rec["trees"].append(row)
            print(json.dumps({"scene": name, **row}), flush=True)
        out.append(rec)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
