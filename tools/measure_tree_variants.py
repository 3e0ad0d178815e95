#!/usr/bin/env python3
"""tools/measure_tree_variants.py (a measurement script: the oracle as a counter; not a test, not product code) — what a better
tree, a wider node or another child order would buy k_trace on BASELINE configs[2], counted by the CPU oracle on the rays of
a pseudo-random texel subset BEFORE anything is built for the GPU (task 5 (b) and (c)).

k_trace is VALU-issue bound (profiles/k_trace_counters.json): what matters is box tests + node overhead + triangle tests
per ray, not bytes.  A quantised 4-ary visit costs ~70 VALU instructions (4 x 12 for the boxes + ~22 for fetch address,
validity, ranking, push/pop; aobaker_amd/csrc/trace.hip, ISA of k_trace<1,1,1,11,2,0>), a triangle test ~64.

  python tools/measure_tree_variants.py [--one-in 1024] [--out profiles/r02/tree_variants.json]
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
    ap.add_argument("--one-in", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--atlas", type=int, default=2048)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc
    import bench

    flat = FlatScene(scenes.terrain_scene(grid=args.grid, atlas=args.atlas, seed=3))
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.accumulate, opt.normal_fp16, opt.seed = args.spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, 3
    tris = orc.scene_triangles(flat.desc)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    sub = bench._subsample(surfels, args.one_in)
    rays = orc.generate_rays(sub, None, opt, args.atlas)
    n_rays = int((rays["extra"][:, 1] != 0).sum())
    rec = {"workload": f"BASELINE configs[2]: {len(tris)} triangles, {args.atlas}^2 atlas, {args.spp} spp, any-hit; rays of a pseudo-random 1/{args.one_in} of the texels",
           "rays": n_rays,
           "cost_model": "VALU instructions per ray ~ 70 x (4-ary visits) + 64 x (triangle tests); a 2^L-ary visit ~ 22 + 12 x 2^L "
                         "(+12 cndmask per 8 children for byte planes)"}

    def four_ary(nodes2, label):
        n4, lt = orc.bvh4_build(nodes2, pairs=True)
        _, frame = orc.bvh2q_build(nodes2)
        n4q = orc.bvh4q_build(n4, frame)
        a, h = orc.walk4q(n4q, frame, lt, rays, True)
        miss_rays = a.rays - h.rays
        out = {"inner_visits_per_ray": round(a.inner_visits / a.rays, 3), "triangle_tests_per_ray": round(a.leaf_visits / a.rays, 3),
               "hit_rate": round(a.hits / a.rays, 4),
               "inner_visits_per_hit_ray": round(h.inner_visits / max(1, h.rays), 3),
               "inner_visits_per_miss_ray": round((a.inner_visits - h.inner_visits) / max(1, miss_rays), 3),
               "valu_per_ray_model": round(70 * a.inner_visits / a.rays + 64 * a.leaf_visits / a.rays, 1)}
        rec[label] = out
        return n4, lt, frame, n4q, out

    t = time.time()
    ploc, _ = orc.bvh_build(tris, 1, 1)
    rec["build_s"] = {"ploc_oracle": round(time.time() - t, 1)}
    n4, lt, frame, n4q, base = four_ary(ploc, "ploc_adaptive_4ary (the default)")

    # (b) tree quality: full top-down binned SAH — the ceiling of any treelet restructuring of the PLOC tree
    t = time.time()
    sah, sah_depth = orc.bvh_build_sah(tris)
    rec["build_s"]["sah_oracle"] = round(time.time() - t, 1)
    _, _, _, _, s = four_ary(sah, "binned_sah_topdown_4ary")
    s["depth"] = sah_depth
    s["vs_default"] = round(s["valu_per_ray_model"] / base["valu_per_ray_model"], 4)
    lbvh, _ = orc.bvh_build(tris, 0, 1)
    _, _, _, _, l = four_ary(lbvh, "lbvh_adaptive_4ary")
    l["vs_default"] = round(l["valu_per_ray_model"] / base["valu_per_ray_model"], 4)

    # (c) node width on the default tree (fp32 boxes; quantisation adds ~0.3 % visits to each alike)
    widths = {}
    for levels in (1, 2, 3, 4):
        st, boxes = orc.kary_visits(ploc, rays, levels, True, True)
        k = 2 ** levels
        per_visit = 22 + 12 * k + (12 * k / 8 if k >= 8 else 0)
        widths[f"{k}-ary"] = {"node_visits_per_ray": round(st.inner_visits / st.rays, 3), "box_tests_per_ray": round(boxes / st.rays, 2),
                              "children_per_visited_node": round(boxes / max(1, st.inner_visits), 2),
                              "triangle_tests_per_ray": round(st.leaf_visits / st.rays, 3),
                              "valu_per_ray_model": round(per_visit * st.inner_visits / st.rays + 64 * st.leaf_visits / st.rays, 1)}
    for k in widths:
        widths[k]["vs_4-ary"] = round(widths[k]["valu_per_ray_model"] / widths["4-ary"]["valu_per_ray_model"], 4)
    rec["node_width_on_default_tree"] = widths

    # child order in any-hit mode (the kernel follows hit children in SLOT order): does a static order find occluders sooner?
    orders = {}
    w = n4q.reshape(-1, 4, 4).copy()
    refs = w[:, :, 3]
    valid = refs != 0xFFFFFFFF
    q = w[:, :, :3] & 0x7fff7fff
    ext = ((q >> 16) - (q & 0xffff)).astype(np.float64) * frame[3:].astype(np.float64)
    area = ext[:, :, 0] * ext[:, :, 1] + ext[:, :, 1] * ext[:, :, 2] + ext[:, :, 2] * ext[:, :, 0]
    top_y = ((q[:, :, 1] >> 16)).astype(np.float64)
    for label, key in (("largest_box_first", -area), ("smallest_box_first", area), ("highest_box_first", -top_y), ("leaves_first", (refs < 0x80000000).astype(np.float64))):
        kk = np.where(valid, key, np.inf)
        perm = np.argsort(kk, axis=1, kind="stable")
        ww = np.take_along_axis(w, perm[:, :, None], axis=1)
        a, h = orc.walk4q(ww.reshape(-1, 16), frame, lt, rays, True)
        orders[label] = {"inner_visits_per_ray": round(a.inner_visits / a.rays, 3), "triangle_tests_per_ray": round(a.leaf_visits / a.rays, 3),
                         "vs_default": round((70 * a.inner_visits + 64 * a.leaf_visits) / a.rays / base["valu_per_ray_model"], 4)}
        assert a.hits == round(base["hit_rate"] * a.rays) or abs(a.hits / a.rays - base["hit_rate"]) < 1e-4
    rec["any_hit_child_order"] = orders
    text = json.dumps(rec, indent=1)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
