#!/usr/bin/env python3
"""tools/measure_two_level_vs_flat.py (a measurement script: the oracle as a counter; not a test, not product code) — node visits per ray of the flat tree vs the two-level (instanced) structure on BASELINE
configs[4], counted by the CPU oracle (no GPU): the measurement behind DESIGN.md §8's "instancing is flattened".

Flat: one BVH over the 10 035 202 world-space triangles (what the library builds; what the reference itself does, since it
never calls CreateInstance).  Two-level: a top-level tree over the 101 instances' world boxes + ONE shared 100 352-triangle
tree for the rocks (+ the ground's), the ray transformed into object space on entry
(kernels/CL/intersect_bvh2level_skiplinks.cl:105-131,225-256).  Same builder (PLOC on the adaptive Morton order), same
traversal discipline (traverse()), same rays: the COSINE sampler's rays of a pseudo-random subset of the 8192^2 atlas,
max distance = falloff distance = 1.25 (configs[4]: "max-ray-distance + cosine falloff on").

  python tools/measure_two_level_vs_flat.py [--spp 64] [--one-in 16384] [--out profiles/r02/two_level_vs_flat.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--one-in", type=int, default=16384)
    ap.add_argument("--instances", type=int, default=100)
    ap.add_argument("--rock-n", type=int, default=224)
    ap.add_argument("--atlas", type=int, default=8192)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc
    import bench

    sc = scenes.instanced_scene(instances=args.instances, rock_n=args.rock_n, atlas=args.atlas, seed=5, bpc=4)
    flat = FlatScene(sc)
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.seed, opt.normal_fp16 = args.spp, capi.SAMPLER_COSINE, 5, capi.FP16_OFF
    opt.use_any_hit, opt.falloff_distance, opt.clamp_tmax_to_falloff = 0, 1.25, 1
    t = time.time()
    nodes, depth = orc.bvh_build(orc.scene_triangles(flat.desc))
    t_flat = time.time() - t
    t = time.time()
    tlas, instances, blas = orc.two_level_scene(sc)
    t_two = time.time() - t
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    sub = bench._subsample(surfels, args.one_in)
    rays = orc.generate_rays(sub, None, opt, args.atlas, t_max=1.25)
    rec = {"workload": f"BASELINE configs[4]: {flat.triangle_count()} triangles ({args.instances} x {2 * args.rock_n ** 2} + ground), {args.atlas}^2 atlas, "
                       f"cosine sampler, max distance 1.25; rays of a pseudo-random 1/{args.one_in} of the {len(surfels)} surfels x {args.spp} spp",
           "rays": int((rays["extra"][:, 1] != 0).sum()),
           "flat": {"nodes": int(len(nodes)), "node_bytes_64": int(len(nodes)) * 64, "depth": int(depth), "oracle_build_s": round(t_flat, 1)},
           "two_level": {"top_nodes": int(len(tlas)), "instances": int(len(instances)), "bottom_trees": len(blas),
                         "bottom_nodes": [int(len(b)) for b in blas], "node_bytes_64": (int(len(tlas)) + sum(len(b) for b in blas)) * 64,
                         "oracle_build_s": round(t_two, 1)}}
    for label, any_hit in (("closest_hit", False), ("any_hit", True)):
        fh, fst = orc.flat_intersect(nodes, rays, any_hit)
        th, top, bottom = orc.two_level_intersect(tlas, instances, blas, rays, any_hit)
        n = fst.rays
        agree = float(((fh["shape_id"] != -1) == (th["shape_id"] != -1)).mean())
        rec[label] = {
            "flat_inner_visits_per_ray": round(fst.inner_visits / n, 3), "flat_leaf_visits_per_ray": round(fst.leaf_visits / n, 3),
            "two_level_top_inner_visits_per_ray": round(top.inner_visits / n, 3),
            "two_level_instances_entered_per_ray": round(top.leaf_visits / n, 3),
            "two_level_bottom_inner_visits_per_ray": round(bottom.inner_visits / n, 3),
            "two_level_leaf_visits_per_ray": round(bottom.leaf_visits / n, 3),
            "two_level_node_visits_over_flat": round((top.inner_visits + bottom.inner_visits + bottom.leaf_visits) / (fst.inner_visits + fst.leaf_visits), 3),
            "hit_rate": round(fst.hits / n, 4), "hit_miss_agreement": round(agree, 6)}
    text = json.dumps(rec, indent=1)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
