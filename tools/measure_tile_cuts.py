#!/usr/bin/env python3
"""tools/measure_tile_cuts.py (a measurement script: the oracle as a counter; not a test, not product code) — what a per-tile
CUT of the tree would take off k_trace's walk, counted by the CPU oracle before the kernel is built.

A tile's rays all start inside one chain of nodes.  A cut (oracle `orc_tile_cut`) replaces, breadth first from the root's four
slots, every inner slot whose box meets / contains the tile's box by its own slots while the list fits E entries: the chain's
SIBLINGS plus whatever the chain ends in.  A new ray tests the E boxes when it is generated (all 64 lanes busy, boxes in scalar
registers) and walks only the subtrees it enters.  Counted: inner and leaf visits of the plain walk (all of them by vector or scalar
visits today) and below the cut, entries per tile, entries entered per ray; the same hits (asserted).

  python tools/measure_tile_cuts.py [--tiles 100] [--entries 32] [--rule 1] [--scene terrain|sphere] [--out FILE]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OrcEntryStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "hits", "inner_visits", "leaf_visits", "entries_tested", "mask_bits")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=100)
    ap.add_argument("--entries", default="16,24,32")
    ap.add_argument("--rule", default="0,1")
    ap.add_argument("--scene", default="terrain")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--atlas", type=int, default=2048)
    ap.add_argument("--builder", type=int, default=2)
    ap.add_argument("--margin", type=float, default=None, help="tile box margin, may be negative (default: |ray bias| + 1e-5)")
    ap.add_argument("--shrink", type=float, default=0.0, help="the tile box pulled in by this fraction of its extent on every side (k_tile_cuts: kCutShrink)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc

    if args.scene == "terrain":
        flat = FlatScene(scenes.terrain_scene(grid=args.grid, atlas=args.atlas, seed=3))
        seed = 3
    else:
        flat = FlatScene(scenes.sphere_scene(n=187, atlas=args.atlas, seed=2))
        seed = 2
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.accumulate, opt.normal_fp16, opt.seed = args.spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, seed
    tris = orc.scene_triangles(flat.desc)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    nodes2 = orc.bvh_build(tris, args.builder, 1)[0]
    n4, lt = orc.bvh4_build(nodes2, pairs=True)
    _, frame = orc.bvh2q_build(nodes2)
    n4q = orc.bvh4q_build(n4, frame)
    lib = orc.load()
    lib.orc_tile_cut.restype = C.c_int32
    lib.orc_tile_cut.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.orc_cut_walk.restype = C.c_int
    lib.orc_cut_walk.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.POINTER(OrcEntryStats), C.c_void_p, C.c_void_p]
    n4q_c, frame_c, lt_c = np.ascontiguousarray(n4q, dtype=np.uint32), np.ascontiguousarray(frame, dtype=np.float32), np.ascontiguousarray(lt)

    # the tiles of the trace order: 8x8 texels where the atlas is fully covered, else 64 consecutive surfels of the row-major list
    rng = np.random.default_rng(7)
    if len(surfels) == args.atlas * args.atlas:
        tw = args.atlas // 8
        grid = surfels.reshape(args.atlas, args.atlas)
        picks = rng.choice(tw * tw, size=args.tiles, replace=False)
        tiles = [np.ascontiguousarray(grid[8 * (t // tw):8 * (t // tw) + 8, 8 * (t % tw):8 * (t % tw) + 8].reshape(-1)) for t in picks]
    else:
        # order the surfels as raster.hip k_tile_* does: by 8x8 texel tile, row-major inside
        xy = surfels["texel"].astype(np.int64) if "texel" in surfels.dtype.names else None
        if xy is None:
            raise SystemExit("surfel records carry no texel field")
        key = ((xy[:, 1] // 8) * (args.atlas // 8) + xy[:, 0] // 8) * 64 + (xy[:, 1] % 8) * 8 + xy[:, 0] % 8
        order = np.argsort(key, kind="stable")
        s2 = surfels[order]
        nt = len(s2) // 64
        picks = rng.choice(nt, size=min(args.tiles, nt), replace=False)
        tiles = [np.ascontiguousarray(s2[64 * t:64 * t + 64]) for t in picks]

    plain = orc.OrcTraceStats()
    ray_sets = []
    for sub in tiles:
        rays = np.ascontiguousarray(orc.generate_rays(sub, None, opt, args.atlas))
        a, _ = orc.walk4q(n4q, frame, lt, rays, True)
        plain.rays += a.rays; plain.inner_visits += a.inner_visits; plain.leaf_visits += a.leaf_visits; plain.hits += a.hits
        ray_sets.append(rays)
    r = float(plain.rays)
    rec = {"workload": f"{args.scene}: {len(tris)} triangles, {args.atlas}^2 atlas, {args.spp} spp, any hit, builder {args.builder}; all rays of {len(tiles)} random tiles",
           "rays": int(r), "hit_rate": round(plain.hits / r, 4),
           "plain_walk": {"inner_visits_per_ray": round(plain.inner_visits / r, 3), "triangle_tests_per_ray": round(plain.leaf_visits / r, 3)},
           "cuts": []}
    margin = abs(opt.ray_bias) + 1e-5 if args.margin is None else args.margin
    for rule in [int(x) for x in args.rule.split(",")]:
        for E in [int(x) for x in args.entries.split(",")]:
            es = OrcEntryStats()
            lengths = []
            cut = np.zeros(4 * 64, dtype=np.uint32)
            for sub, rays in zip(tiles, ray_sets):
                pos = sub["position"].astype(np.float32)
                pmin, pmax = pos.min(axis=0), pos.max(axis=0)
                pull = np.float32(args.shrink) * (pmax - pmin)
                lo = np.ascontiguousarray((pmin + pull) - np.float32(margin), dtype=np.float32)
                hi = np.ascontiguousarray((pmax - pull) + np.float32(margin), dtype=np.float32)
                m = lib.orc_tile_cut(n4q_c.ctypes.data, len(n4q_c), frame_c.ctypes.data, lo.ctypes.data, hi.ctypes.data, E, rule, cut.ctypes.data)
                lengths.append(m)
                rc = lib.orc_cut_walk(n4q_c.ctypes.data, len(n4q_c), frame_c.ctypes.data, lt_c.ctypes.data, rays.ctypes.data, len(rays), cut.ctypes.data, m, C.byref(es), None, None)
                assert rc == 0
            assert es.hits == plain.hits and es.rays == plain.rays, (es.hits, plain.hits)
            rec["cuts"].append({"rule": "meets" if rule else "contains", "max_entries": E,
                                "entries_per_tile_mean": round(float(np.mean(lengths)), 2), "entries_per_tile_min_max": [int(min(lengths)), int(max(lengths))],
                                "entries_entered_per_ray": round(es.mask_bits / r, 3),
                                "inner_visits_per_ray_below_the_cut": round(es.inner_visits / r, 3),
                                "triangle_tests_per_ray": round(es.leaf_visits / r, 3)})
            print(json.dumps(rec["cuts"][-1]), flush=True)
    text = json.dumps(rec, indent=1)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
