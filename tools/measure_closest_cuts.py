#!/usr/bin/env python3
"""tools/measure_closest_cuts.py (a measurement script: the oracle as a counter; not a test, not product code) — what a per-tile CUT
of the tree takes off the CLOSEST-HIT walk of short rays (BASELINE configs[4]: falloff scoring, tmax clamped to 1.25), counted by the
CPU oracle before the kernel is built (round 5, VERDICT r4 task 2).

The cut is orc_tile_cut_reach: k_tile_cuts' rule, and slots farther than the rays' reach from the tile's box never join the list.
Counted per ray: inner visits and triangle tests of the walk from the root (traverse4q, closest hit) and below the cut, entries tested
and entered — for both orders of the entered entries and with / without a second sweep that spends the budget the dropped siblings
free on the tree around the tile.  Every ray's closest distance must equal the root walk's (asserted).

  python tools/measure_closest_cuts.py [--instances 100] [--rock 64] [--atlas 2048] [--spp 64] [--tiles 60] [--out FILE]
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
    ap.add_argument("--scene", default="instanced")
    ap.add_argument("--instances", type=int, default=100)
    ap.add_argument("--rock", type=int, default=64)
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--atlas", type=int, default=2048)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--tiles", type=int, default=60)
    ap.add_argument("--entries", default="32")
    ap.add_argument("--shrink", type=float, default=0.25)
    ap.add_argument("--grow2", default="-1,0,0.1,0.25,0.5")
    ap.add_argument("--builder", type=int, default=1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc

    if args.scene == "instanced":
        flat = FlatScene(scenes.instanced_scene(instances=args.instances, rock_n=args.rock, atlas=args.atlas, seed=5, bpc=1))
        seed = 5
    else:
        flat = FlatScene(scenes.terrain_scene(grid=args.grid, atlas=args.atlas, seed=3))
        seed = 3
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.accumulate, opt.normal_fp16, opt.seed = args.spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, seed
    opt.use_any_hit, opt.falloff_distance, opt.clamp_tmax_to_falloff = 0, 1.25, 1
    t_max = np.float32(1.25)
    tris = orc.scene_triangles(flat.desc)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    nodes2 = orc.bvh_build(tris, args.builder, 1)[0]
    n4, lt = orc.bvh4_build(nodes2, pairs=True)
    _, frame = orc.bvh2q_build(nodes2)
    n4q = orc.bvh4q_build(n4, frame)
    lib = orc.load()
    lib.orc_tile_cut_reach.restype = C.c_int32
    lib.orc_tile_cut_reach.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_float, C.c_void_p]
    lib.orc_cut_walk_closest.restype = C.c_int
    lib.orc_cut_walk_closest.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(OrcEntryStats), C.c_void_p]
    lib.orc_walk4q_t.restype = C.c_int
    lib.orc_walk4q_t.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]
    n4q_c, frame_c, lt_c = np.ascontiguousarray(n4q, dtype=np.uint32), np.ascontiguousarray(frame, dtype=np.float32), np.ascontiguousarray(lt)

    # the tiles of the trace order (raster.hip k_tile_*): by 8x8 texel tile, row-major inside
    xy = np.stack([surfels["texel_x"], surfels["texel_y"]], axis=1).astype(np.int64) if "texel_x" in surfels.dtype.names else surfels["texel"].astype(np.int64)
    key = ((xy[:, 1] // 8) * (args.atlas // 8) + xy[:, 0] // 8) * 64 + (xy[:, 1] % 8) * 8 + xy[:, 0] % 8
    s2 = surfels[np.argsort(key, kind="stable")]
    nt = len(s2) // 64
    rng = np.random.default_rng(7)
    picks = rng.choice(nt, size=min(args.tiles, nt), replace=False)
    tiles = [np.ascontiguousarray(s2[64 * t:64 * t + 64]) for t in picks]

    plain = orc.OrcTraceStats()
    ray_sets, want_t = [], []
    for sub in tiles:
        rays = np.ascontiguousarray(orc.generate_rays(sub, None, opt, args.atlas, t_max=float(t_max)))
        a = orc.OrcTraceStats()
        t = np.zeros(len(rays), dtype=np.float32)
        assert lib.orc_walk4q_t(n4q_c.ctypes.data, len(n4q_c), frame_c.ctypes.data, lt_c.ctypes.data, rays.ctypes.data, len(rays), 0, C.byref(a), t.ctypes.data) == 0
        plain.rays += a.rays; plain.inner_visits += a.inner_visits; plain.leaf_visits += a.leaf_visits; plain.hits += a.hits
        ray_sets.append(rays); want_t.append(t)
    r = float(plain.rays)
    # (measurement: the same rays following their hit children in slot order — the any-hit visit, no ranking — with closest-hit pruning)
    lib.orc_measure_slot_order(1)
    slot = orc.OrcTraceStats()
    for rays, wt in zip(ray_sets, want_t):
        t = np.zeros(len(rays), dtype=np.float32)
        assert lib.orc_walk4q_t(n4q_c.ctypes.data, len(n4q_c), frame_c.ctypes.data, lt_c.ctypes.data, rays.ctypes.data, len(rays), 0, C.byref(slot), t.ctypes.data) == 0
        assert np.array_equal(t, wt)
    lib.orc_measure_slot_order(0)
    print(json.dumps({"slot_order_from_root": {"inner_visits_per_ray": round(slot.inner_visits / r, 3), "triangle_tests_per_ray": round(slot.leaf_visits / r, 3)}, "hit_rate": round(plain.hits / r, 4)}), flush=True)
    rec = {"workload": f"{args.scene}: {len(tris)} triangles, {args.atlas}^2 atlas, {args.spp} spp, closest hit, tmax = falloff = 1.25, builder {args.builder}; all rays of {len(tiles)} random tiles",
           "rays": int(r), "hit_rate": round(plain.hits / r, 4),
           "walk_from_the_root": {"inner_visits_per_ray": round(plain.inner_visits / r, 3), "triangle_tests_per_ray": round(plain.leaf_visits / r, 3)},
           "cuts": []}
    print(json.dumps(rec["walk_from_the_root"]), flush=True)
    reach0 = float((np.float32(t_max) + np.float32(abs(opt.ray_bias))) * np.float32(1.01))
    for E in [int(x) for x in args.entries.split(",")]:
        for reach in (0.0, reach0):
            for grow2 in ([float(x) for x in args.grow2.split(",")] if reach > 0 else [-1.0]):
                for order in ((0, 1) if reach == 0 else (1, 2, 3)):
                    es = OrcEntryStats()
                    lengths = []
                    cut = np.zeros(4 * 128, dtype=np.uint32)
                    for sub, rays, wt in zip(tiles, ray_sets, want_t):
                        pos = sub["position"].astype(np.float32)
                        pmin, pmax = pos.min(axis=0), pos.max(axis=0)
                        pull = np.float32(args.shrink) * (pmax - pmin)
                        lo, hi = np.ascontiguousarray(pmin + pull, dtype=np.float32), np.ascontiguousarray(pmax - pull, dtype=np.float32)
                        flo, fhi = np.ascontiguousarray(pmin), np.ascontiguousarray(pmax)
                        m = lib.orc_tile_cut_reach(n4q_c.ctypes.data, len(n4q_c), frame_c.ctypes.data, lo.ctypes.data, hi.ctypes.data, flo.ctypes.data, fhi.ctypes.data,
                                                   reach, E, grow2, cut.ctypes.data)
                        lengths.append(m)
                        t_out = np.zeros(len(rays), dtype=np.float32)
                        rc = lib.orc_cut_walk_closest(n4q_c.ctypes.data, len(n4q_c), frame_c.ctypes.data, lt_c.ctypes.data, rays.ctypes.data, len(rays), cut.ctypes.data, m, order,
                                                      C.byref(es), t_out.ctypes.data)
                        assert rc == 0
                        assert np.array_equal(t_out, wt), "the closest distances differ from the root walk's"
                    assert es.hits == plain.hits and es.rays == plain.rays, (es.hits, plain.hits)
                    rec["cuts"].append({"max_entries": E, "reach": round(reach, 4), "second_sweep_grow": grow2, "order": ("shallowest first", "deepest first", "nearest first (by the entry's near distance)", "deepest first, the entry's box tested again")[order],
                                        "entries_per_tile_mean": round(float(np.mean(lengths)), 2), "entries_per_tile_min_max": [int(min(lengths)), int(max(lengths))],
                                        "entries_entered_per_ray": round(es.mask_bits / r, 3),
                                        "inner_visits_per_ray_below_the_cut": round(es.inner_visits / r, 3),
                                        "triangle_tests_per_ray": round(es.leaf_visits / r, 3)})
                    print(json.dumps(rec["cuts"][-1]), flush=True)
    text = json.dumps(rec, indent=1)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
