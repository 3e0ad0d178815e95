#!/usr/bin/env python3
"""tools/measure_far_tables.py (a measurement script, not a test, not product code) — what a SECOND horizon table per surfel, built from the far field alone, would buy.

The product's tables (k_horizon) prove cells of the sampler's grid free of ALL geometry: their samples are not cast.  A table built from the geometry that does not lie
wholly within R of the tile (the same kernel, every box wholly within R ignored: tests/horizon_emu, `far_from`) proves a cell free of everything a ray could hit beyond R:
its samples could be cast with tmax clipped to R — the same answer (what lies wholly within R is all such a ray can hit), far fewer nodes.  Here, on the CPU: the kernel's
own logic for both tables on random tiles of BASELINE configs[2], the oracle's rays and its walk below the tiles' cuts for the visits.
  python tools/measure_far_tables.py [--tiles 12] [--radii 2,4,8]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=12)
    ap.add_argument("--radii", default="2,4,8")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import horizon_emu as he
    from aobaker_amd import capi
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc
    flat = FlatScene(he.scene_of("terrain", 2048))
    opt = orc.default_options()
    spp = a.spp
    opt.sample_count, opt.sampler, opt.accumulate, opt.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, 3
    tris = orc.scene_triangles(flat.desc)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    nodes2 = orc.bvh_build(tris, 2, 1)[0]
    n4, lt = orc.bvh4_build(nodes2, pairs=True)
    _, qframe = orc.bvh2q_build(nodes2)
    n4q = orc.bvh4q_build(n4, qframe)
    W, H = flat.material_array[0].width, flat.material_array[0].height
    surfels = he.tile_order(surfels.reshape(H, W))
    ntiles = len(surfels) // 64
    rng = np.random.default_rng(a.seed)
    picks = np.sort(rng.choice(ntiles, size=a.tiles, replace=False))
    sub = np.concatenate([surfels[64 * t:64 * t + 64] for t in picks])
    sx = int(np.floor(np.sqrt(spp))); sy = spp // sx
    lib = he.load()
    lib.emu_set_far_from.argtypes = [__import__("ctypes").c_float]
    lib.emu_set_far_from(0.0)
    tab_all = he.tables(sub, n4q, lt, qframe, spp, bias=float(opt.ray_bias))[:, :sy].astype(np.int64)
    rays = np.ascontiguousarray(orc.generate_rays(sub, None, opt, W))
    occ = (orc.occluded_rays(nodes2, rays) == 1)
    cast = ~(np.arange(sx)[None, None, :] < tab_all[:, :, None]).reshape(-1)
    assert not (occ & ~cast).any()
    # the tiles' cuts and the walk below them
    def walk(sel_rays_by_tile):
        tot = {"rays": 0, "inner": 0, "leaf": 0, "hits": 0}
        for ti, rr in sel_rays_by_tile:
            if len(rr) == 0:
                continue
            P = sub[64 * ti:64 * ti + 64]["position"].astype(np.float32)
            lo, hi = P.min(0), P.max(0)
            shrink = 0.25 * (hi - lo)
            cut = orc.tile_cut(n4q, qframe, lo + shrink, hi - shrink)
            st, hits = orc.cut_walk(n4q, qframe, lt, rr, cut)
            tot["rays"] += int(st.rays); tot["inner"] += int(st.inner_visits); tot["leaf"] += int(st.leaf_visits); tot["hits"] += int(hits.sum())
        return tot
    tmax_field = [n for n in rays.dtype.names if "max" in n.lower() or n == "o"]
    out = {"workload": f"BASELINE configs[2], {a.tiles} random tiles of 64 surfels, {spp} spp, builder 2; tables by the kernel's own logic on the CPU (tests/horizon_emu)",
           "rays": int(len(rays)), "occluded_share": round(float(occ.mean()), 4), "cast_share_with_the_product_tables": round(float(cast.mean()), 4)}
    per_tile = lambda mask, rs: [(ti, rs[ti * 64 * spp:(ti + 1) * 64 * spp][mask[ti * 64 * spp:(ti + 1) * 64 * spp]]) for ti in range(a.tiles)]
    base = walk(per_tile(cast, rays))
    out["cast_rays_walk"] = {**base, "inner_per_ray": round(base["inner"] / max(1, base["rays"]), 2), "leaf_per_ray": round(base["leaf"] / max(1, base["rays"]), 2)}
    diag = np.array([np.linalg.norm(sub[64 * t:64 * t + 64]["position"].max(0) - sub[64 * t:64 * t + 64]["position"].min(0)) for t in range(a.tiles)])
    out["radii"] = []
    for R in [float(x) for x in a.radii.split(",")]:
        lib.emu_set_far_from(R)
        cnt = np.zeros(8, np.uint64)
        tab_far = he.tables(sub, n4q, lt, qframe, spp, bias=float(opt.ray_bias), counters=cnt)[:, :sy].astype(np.int64)
        lib.emu_set_far_from(0.0)
        farfree = (np.arange(sx)[None, None, :] < tab_far[:, :, None]).reshape(-1)
        short = cast & farfree
        clipped = rays.copy()
        # the ray record's reach: o.w (RadeonRays ray: o.xyz, max t in o.w)
        tm = np.repeat(R * diag, 64 * spp).astype(np.float32)
        o = clipped["o"] if "o" in clipped.dtype.names else None
        if o is None:
            raise SystemExit("ray dtype: " + str(clipped.dtype))
        o[:, 3] = np.minimum(o[:, 3], tm)
        w_short_full = walk(per_tile(short, rays))
        w_short_clip = walk(per_tile(short, clipped))
        # soundness on these rays: the clipped walk finds a hit exactly where the full walk does
        T = float(max(1, int(cnt[5])))
        out["radii"].append({"R_tile_diagonals": R, "far_free_share_of_all_samples": round(float(farfree.mean()), 4), "short_share_of_all_samples": round(float(short.mean()), 4),
                             "short_share_of_cast": round(float(short.sum() / max(1, cast.sum())), 4),
                             "short_rays_inner_visits_full_tmax": w_short_full["inner"], "short_rays_inner_visits_clipped": w_short_clip["inner"],
                             "short_rays_hits_full": w_short_full["hits"], "short_rays_hits_clipped": w_short_clip["hits"],
                             "inner_visits_of_all_cast_rays": base["inner"],
                             "inner_visits_saved_share": round((w_short_full["inner"] - w_short_clip["inner"]) / max(1, base["inner"]), 4),
                             "far_pass_per_tile": {"rounds": round(int(cnt[0]) / T, 1), "box_tests": round(int(cnt[1]) / T, 1), "near_triangles": round(int(cnt[2]) / T, 1), "middle_leaves": round(int(cnt[3]) / T, 1)}})
        print(json.dumps(out["radii"][-1]), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
