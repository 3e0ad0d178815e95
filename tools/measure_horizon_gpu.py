#!/usr/bin/env python3
"""tools/measure_horizon_gpu.py — GPU check of the horizon tables (horizon.hip) against the oracle, outside pytest: for a list of scenes, the cells baked
with tables == the cells baked without == the oracle's; every cell the tables skip holds only unoccluded rays of the oracle; the share skipped.
  python tools/measure_horizon_gpu.py [--big]
"""
import argparse
import json
import sys
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.bake_engine import Baker
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc
    lib = capi.load()
    cfg = capi.AobDeviceConfig(); cfg.device = 0
    capi.check(lib.aob_init(cfg))
    cases = [("cube", scenes.cube_scene(64, bpc=4), 16, True), ("cube256", scenes.cube_scene(256, bpc=4), 64, True),
             ("inward", scenes.cube_scene(64, inward=True, bpc=4), 64, True),
             ("sphere", scenes.sphere_scene(n=40, atlas=128), 64, True), ("soup", scenes.soup_scene(atlas=96), 64, True), ("floor_wall", scenes.floor_wall_scene(atlas=32), 256, True), ("two_section", scenes.two_section_scene(atlas=32), 64, True),
             ("terrain_small", scenes.terrain_scene(grid=64, atlas=256, seed=3), 256, True),
             ("terrain_small_closest", scenes.terrain_scene(grid=64, atlas=256, seed=3), 256, False)]
    if args.big:
        cases.append(("terrain 1M 2048 x 256", scenes.terrain_scene(grid=708, atlas=2048, seed=3), 256, True))
    for name, scene, spp, anyhit in cases:
        if scene is None:
            continue
        flat = FlatScene(scene)
        res = {}
        for hz in (1, 0):
            opt = capi.default_options()
            opt.sample_count, opt.sampler, opt.accumulate, opt.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, 3
            opt.use_any_hit = 1 if anyhit else 0
            opt.trace_horizon = hz
            with Baker(flat, opt) as bk:
                bk.build_bvh(); bk.rasterize(0); bk.trace(0); bk.sync()
                t0 = time.time(); bk.clear_results(0); bk.build_bvh(); bk.rasterize(0); bk.trace(0); bk.sync(); dt = time.time() - t0
                st = bk.stats()
                res[hz] = (bk.results(0).copy(), st.ms_trace, st.rays_cast, st.ray_count, st.trace_horizon, bk.horizon(0) if hz == 0 else None, bk.surfels(0))
        same = bool((res[0][0]["total_bits"] == res[1][0]["total_bits"]).all() and (res[0][0]["weight_bits"] == res[1][0]["weight_bits"]).all())
        rec = {"scene": name, "spp": spp, "any_hit": anyhit, "cells_equal_with_and_without_tables": same, "trace_ms_with": round(res[0][1], 3), "trace_ms_without": round(res[1][1], 3),
               "rays": int(res[0][3]), "rays_cast": int(res[0][2]), "cast_share": round(res[0][2] / max(1, res[0][3]), 4), "trace_horizon": int(res[0][4])}
        if not args.no_oracle and res[0][3] <= (1 << 25):
            # the oracle's rays: every skipped cell must hold unoccluded rays only
            hzt, surfels = res[0][5], res[0][6]
            if hzt is not None and len(hzt):
                tris = orc.scene_triangles(flat.desc)
                nodes2 = orc.bvh_build(tris, 1, 1)[0]
                sx = int(np.floor(np.sqrt(spp))); sy = spp // sx
                opt = capi.default_options()
                opt.sample_count, opt.sampler, opt.accumulate, opt.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, 3
                W = flat.material_array[0].width
                rays = orc.generate_rays(surfels, None, opt, W)
                occ = (orc.occluded_rays(nodes2, rays).reshape(len(surfels), spp) == 1).reshape(len(surfels), sy, sx)
                free = hzt[:, :sy].astype(np.int64)
                skipped = np.arange(sx)[None, None, :] < free[:, :, None]
                rec["skipped_cells_with_an_occluded_ray"] = int((skipped & occ).sum())
                rec["skipped_share_by_table"] = round(float(skipped.mean()), 4)
                rec["occluded_share"] = round(float(occ.mean()), 4)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
