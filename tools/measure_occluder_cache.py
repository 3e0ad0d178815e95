#!/usr/bin/env python3
"""tools/measure_occluder_cache.py (a measurement script: the oracle as a counter; not a test, not product code) — what an OCCLUDER
CACHE would take off the any-hit walk of BASELINE configs[2], counted before anything is built (task 1c).

Idea: before its walk an any-hit ray tests the grid cell (the triangle pair) that ended an earlier occluded ray — if it hits, the
ray is occluded and never walks; an any-hit answer does not depend on WHICH occluder is found.  Counted on the oracle's own rays
and closest-hit records (orc.intersect_rays: the occluder a ray would most likely find) for two cache keys:
  surfel   the pair that occluded the surfel's previous occluded sample (samples in generation order);
  stratum  the pair that occluded the SAME sample of the previous surfel of the tile (a neighbouring surfel, the same stratum cell).
Reported: the share of occluded rays the cached pair also occludes (exact Möller–Trumbore in float64 on the cached pair), the node
visits a cache hit saves (the oracle's walk of those rays) and what the test costs (one pair test per ray).

  python tools/measure_occluder_cache.py [--tiles 40] [--out FILE]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hits_pair(o, d, tri, tmax):
    """Does ray (o, d) hit triangle tri [3, 3] within (0, tmax)?  float64 Möller–Trumbore."""
    e1, e2 = tri[1] - tri[0], tri[2] - tri[0]
    s1 = np.cross(d, e2)
    den = s1 @ e1
    if den == 0.0:
        return False
    inv = 1.0 / den
    dd = o - tri[0]
    b1 = (dd @ s1) * inv
    s2 = np.cross(dd, e1)
    b2 = (d @ s2) * inv
    t = (e2 @ s2) * inv
    return b1 >= 0 and b2 >= 0 and b1 + b2 <= 1 and 0 <= t <= tmax


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=40)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc
    atlas, spp = 2048, 256
    flat = FlatScene(scenes.terrain_scene(grid=708, atlas=atlas, seed=3))
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.accumulate, opt.normal_fp16, opt.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, 3
    tris = orc.scene_triangles(flat.desc)
    verts = tris["v"].astype(np.float64)               # [n, 3, 3], indexed by prim_id (one mesh: the scene order)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    nodes2 = orc.bvh_build(tris, 2, 1)[0]
    n4, lt = orc.bvh4_build(nodes2, pairs=True)
    _, frame = orc.bvh2q_build(nodes2)
    n4q = orc.bvh4q_build(n4, frame)
    rng = np.random.default_rng(7)
    tw = atlas // 8
    grid = surfels.reshape(atlas, atlas)
    tot = {"rays": 0, "occluded": 0, "surfel_hits": 0, "stratum_hits": 0, "surfel_tests": 0, "stratum_tests": 0}
    visits_all = visits_occ = 0
    for t in rng.choice(tw * tw, size=args.tiles, replace=False):
        ty, tx = divmod(int(t), tw)
        sub = np.ascontiguousarray(grid[8 * ty:8 * ty + 8, 8 * tx:8 * tx + 8].reshape(-1))
        rays = np.ascontiguousarray(orc.generate_rays(sub, None, opt, atlas))
        recs = orc.intersect_rays(nodes2, rays)
        prim = recs["prim_id"].reshape(64, spp)
        occ = recs["shape_id"].reshape(64, spp) > 0            # shape ids start at 1; a miss leaves -1
        o = rays["o"][:, :3].astype(np.float64).reshape(64, spp, 3)
        d = rays["d"][:, :3].astype(np.float64).reshape(64, spp, 3)
        tmax = float(rays["o"][0, 3])
        a, h = orc.walk4q(n4q, frame, lt, rays, True)
        visits_all += a.inner_visits + a.leaf_visits
        visits_occ += h.inner_visits + h.leaf_visits
        tot["rays"] += 64 * spp
        tot["occluded"] += int(occ.sum())

        def pair_of(p):            # the two halves of a grid cell are consecutive triangles (aobaker_amd/scenes.py _grid_indices)
            return (p // 2) * 2

        # key = surfel: the previous occluded sample of the same surfel
        for s in range(64):
            cached = -1
            for i in range(spp):
                if cached >= 0:
                    tot["surfel_tests"] += 1
                    if occ[s, i] and (hits_pair(o[s, i], d[s, i], verts[cached], tmax) or hits_pair(o[s, i], d[s, i], verts[cached + 1], tmax)):
                        tot["surfel_hits"] += 1
                if occ[s, i]:
                    cached = pair_of(int(prim[s, i]))
        # key = stratum: the same sample of the previous surfel of the tile
        for i in range(spp):
            cached = -1
            for s in range(64):
                if cached >= 0:
                    tot["stratum_tests"] += 1
                    if occ[s, i] and (hits_pair(o[s, i], d[s, i], verts[cached], tmax) or hits_pair(o[s, i], d[s, i], verts[cached + 1], tmax)):
                        tot["stratum_hits"] += 1
                if occ[s, i]:
                    cached = pair_of(int(prim[s, i]))
    r = float(tot["rays"])
    per_occ = visits_occ / max(1, tot["occluded"])
    rec = {"workload": f"BASELINE configs[2]: all rays of {args.tiles} random 8x8-texel tiles, bvh_builder 2; occluders = the closest-hit records of orc.intersect_rays",
           "rays": int(r), "occluded_share": round(tot["occluded"] / r, 4),
           "node_visits_per_ray_all": round(visits_all / r, 3), "node_visits_per_occluded_ray": round(per_occ, 3),
           "cache_by_surfel": {"hit_share_of_occluded_rays": round(tot["surfel_hits"] / max(1, tot["occluded"]), 4),
                               "visits_saved_per_ray": round(tot["surfel_hits"] * per_occ / r, 3),
                               "saved_share_of_all_visits": round(tot["surfel_hits"] * per_occ / max(1, visits_all), 4)},
           "cache_by_stratum_neighbour": {"hit_share_of_occluded_rays": round(tot["stratum_hits"] / max(1, tot["occluded"]), 4),
                                          "visits_saved_per_ray": round(tot["stratum_hits"] * per_occ / r, 3),
                                          "saved_share_of_all_visits": round(tot["stratum_hits"] * per_occ / max(1, visits_all), 4)},
           "cost": "one triangle-pair test per ray (about one leaf visit's arithmetic) for every ray, occluded or not",
           "rule": "build it only if it removes >= 8 % of the lane-visits (VERDICT r3)"}
    text = json.dumps(rec, indent=1)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
