#!/usr/bin/env python3
"""tools/measure_horizon_cull.py (a measurement script: the oracle as a counter; not a test, not product code) — how many rays of a bake a
per-surfel HORIZON TABLE could resolve without tracing them.

The COSINE sampler is stratified on (u1, u2): u2 picks the azimuth in the surfel's tangent frame, u1 = sin^2(theta) the distance from the normal.
For every surfel and azimuth sector the table holds the largest u1 below which NO geometry can lie: any ray of that sector whose stratum ends below
it is unoccluded whatever its jitter, and need not be generated.  Built conservatively from the triangles themselves: a triangle seen from P in the
frame (X, N, Z) spans an arc of azimuths and lies below beta_max / rho_min in slope, hence blocks only u1 >= rho_min^2 / (rho_min^2 + beta_max^2)
(and, because a hit lies at least `ray_bias` along the ray, u1 >= 1 - (beta_max / bias)^2).

Measures on BASELINE configs[2]: the share of rays culled (a) by this table built from every triangle (the tightest the rule can be), (b) by the
ideal table read off the oracle's own rays, and checks (a) against the oracle: no culled ray may be occluded.

  python tools/measure_horizon_cull.py [--tiles 24] [--surfels 8] [--out profiles/r06/horizon_cull_prototype.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def seg_dist2(ax, az, bx, bz):
    """squared distance from the origin to the segments a-b (vectorised)"""
    dx, dz = bx - ax, bz - az
    l2 = dx * dx + dz * dz
    t = np.where(l2 > 0, -(ax * dx + az * dz) / np.where(l2 > 0, l2, 1.0), 0.0)
    t = np.clip(t, 0.0, 1.0)
    px, pz = ax + t * dx, az + t * dz
    return px * px + pz * pz


def surfel_table(P, N, X, Z, V, sectors, bias, margin_u1=1e-3, margin_phi=1e-3):
    """table[sector] = the largest u1 that is certainly free, from all triangles V (n, 3, 3)"""
    v = V - P
    al = v @ X
    ga = v @ Z
    be = (v @ N) / float(N @ N)
    bmax = be.max(1)
    keep = bmax > -1e-9
    al, ga, bmax = al[keep], ga[keep], np.maximum(bmax[keep], 0.0)
    # the origin inside the projected triangle?
    def cr(i, j):
        return al[:, i] * ga[:, j] - ga[:, i] * al[:, j]
    c0, c1, c2 = cr(0, 1), cr(1, 2), cr(2, 0)
    inside = ((c0 >= 0) & (c1 >= 0) & (c2 >= 0)) | ((c0 <= 0) & (c1 <= 0) & (c2 <= 0))
    d2 = np.minimum(np.minimum(seg_dist2(al[:, 0], ga[:, 0], al[:, 1], ga[:, 1]), seg_dist2(al[:, 1], ga[:, 1], al[:, 2], ga[:, 2])),
                    seg_dist2(al[:, 2], ga[:, 2], al[:, 0], ga[:, 0]))
    d2 = np.where(inside, 0.0, d2)
    u_a = d2 / np.maximum(d2 + bmax * bmax, 1e-300)
    u_b = 1.0 - np.minimum(1.0, (bmax / bias)) ** 2
    u_block = np.maximum(u_a, u_b) - margin_u1
    # arc of azimuths (fractions of a turn)
    th = np.arctan2(ga, al) / (2 * np.pi)
    rel = (th - th[:, :1] + 0.5) % 1.0 - 0.5
    lo = th[:, 0] + rel.min(1) - margin_phi
    hi = th[:, 0] + rel.max(1) + margin_phi
    full = inside | ((hi - lo) >= 0.5)
    table = np.ones(sectors)
    s_lo = np.floor(lo * sectors).astype(np.int64)
    s_hi = np.floor(hi * sectors).astype(np.int64)
    for s in range(sectors):
        # sector s (mod sectors) within [s_lo, s_hi]
        k = np.ceil((s_lo - s) / sectors).astype(np.int64)
        touch = full | (s + k * sectors <= s_hi)
        if touch.any():
            table[s] = min(1.0, max(0.0, u_block[touch].min()))
    return table, int(keep.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=24)
    ap.add_argument("--surfels", type=int, default=8)
    ap.add_argument("--atlas", type=int, default=2048)
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--sectors", type=int, default=16)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc
    atlas = args.atlas
    flat = FlatScene(scenes.terrain_scene(grid=args.grid, atlas=atlas, seed=3))
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.accumulate, opt.seed = args.spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, 3
    tris = orc.scene_triangles(flat.desc)
    V = tris["v"].astype(np.float64)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    nodes2 = orc.bvh_build(tris, 2, 1)[0]
    grid = surfels.reshape(atlas, atlas)
    tw = atlas // 8
    rng = np.random.default_rng(11)
    picks = rng.choice(tw * tw, size=args.tiles, replace=False)
    sx = int(np.floor(np.sqrt(args.spp)))
    sy = args.spp // sx
    bias = float(opt.ray_bias)
    n = dict(rays=0, occluded=0, culled=0, culled_ideal=0, culled_ideal_margin=0, wrong=0, surfels=0, tris_above=0)
    per_col = np.zeros(sx, dtype=np.int64)
    for t_ in picks:
        sub = np.ascontiguousarray(grid[8 * (t_ // tw):8 * (t_ // tw) + 8, 8 * (t_ % tw):8 * (t_ % tw) + 8].reshape(-1))
        rays = np.ascontiguousarray(orc.generate_rays(sub, None, opt, atlas))
        occ = (orc.occluded_rays(nodes2, rays).reshape(64, args.spp) == 1)
        for s_ in rng.choice(64, size=args.surfels, replace=False):
            P = sub["position"][s_].astype(np.float64)
            N = sub["normal"][s_].astype(np.float64)
            if -0.9999 < N[1] < 0.9999:
                X = np.array([-N[2], 0.0, N[0]]); X /= np.linalg.norm(X)
                Z = np.cross(X, N); Z /= np.linalg.norm(Z)
            else:
                X, Z = np.array([1.0, 0, 0]), np.array([0, 0, 1.0])
            table, above = surfel_table(P, N, X, Z, V, args.sectors, bias)
            o = occ[s_][:sx * sy].reshape(sy, sx)                  # [cy][cx]
            cx = np.arange(sx)
            for cy in range(sy):
                # the sectors row cy spans
                a, b = int(np.floor(cy / sy * args.sectors)), int(np.ceil((cy + 1) / sy * args.sectors)) - 1
                free_u1 = min(table[s % args.sectors] for s in range(a, b + 1))
                culled = (cx + 1) / sx <= free_u1
                n["culled"] += int(culled.sum())
                n["wrong"] += int((culled & o[cy]).sum())
                per_col += culled
                first = int(np.argmax(o[cy])) if o[cy].any() else sx
                n["culled_ideal"] += first
                n["culled_ideal_margin"] += max(0, first - 1)
            n["rays"] += sx * sy; n["occluded"] += int(o.sum()); n["surfels"] += 1; n["tris_above"] += above
    r = float(n["rays"])
    rec = {"workload": f"BASELINE configs[2]: terrain {len(tris)} triangles, {atlas}^2, {args.spp} spp; {args.surfels} surfels of each of {args.tiles} random tiles; "
                       f"{args.sectors} azimuth sectors per surfel",
           "rays": n["rays"], "occluded_share": round(n["occluded"] / r, 4),
           "culled_share": {"table from every triangle (conservative rule)": round(n["culled"] / r, 4),
                            "ideal: up to the first occluded stratum of the oracle's own rays": round(n["culled_ideal"] / r, 4),
                            "ideal less one column": round(n["culled_ideal_margin"] / r, 4)},
           "culled_by_column": [round(float(x) / (n["surfels"] * sy), 3) for x in per_col],
           "culled_rays_that_the_oracle_finds_occluded": n["wrong"],
           "triangles_above_the_tangent_plane_per_surfel": round(n["tris_above"] / n["surfels"], 1)}
    text = json.dumps(rec, indent=1)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
