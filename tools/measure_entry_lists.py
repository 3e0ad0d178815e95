#!/usr/bin/env python3
"""tools/measure_entry_lists.py (a measurement script: the oracle as a counter; not a test, not product code) — what a per-tile
ENTRY LIST would take off k_trace's walk on BASELINE configs[2], counted by the CPU oracle before the kernel is built.

The rays of one 8x8-texel tile all start inside the same few top nodes of the tree.  An entry list names up to 8 of those
nodes (breadth first from the root: children whose box overlaps the tile's box); every new ray tests their 4 x 8 child boxes
when it is generated — all 64 lanes busy, the nodes in scalar registers — and starts its walk with a 32-bit mask of the hit
children that are not entries themselves.  Counted here: inner visits of the plain walk, of the walk after the entry phase,
entries tested and mask bits per ray.

  python tools/measure_entry_lists.py [--tiles 200] [--entries 8] [--rule overlap|contain] [--out profiles/r03/entry_lists_prototype.json]
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


def entry_list(n4q, frame, lo, hi, max_entries, rule):
    """breadth first from the root: inner children whose quantised box overlaps (or contains) the box [lo, hi]"""
    origin, scale = frame[:3].astype(np.float64), frame[3:].astype(np.float64)
    glo = np.where(scale > 0, (lo - origin) / np.where(scale > 0, scale, 1), 0.0)
    ghi = np.where(scale > 0, (hi - origin) / np.where(scale > 0, scale, 1), 0.0)
    entries = [0]
    e = 0
    while e < len(entries):
        node = entries[e] & 0x3ffffff
        w = n4q[node].reshape(4, 4)
        for j in range(4):
            ref = int(w[j, 3])
            if ref >= 0x80000000:
                continue
            q = w[j, :3] & 0x7fff7fff
            qmin, qmax = (q & 0xffff).astype(np.float64), (q >> 16).astype(np.float64)
            ok = np.all(qmin <= glo) and np.all(qmax >= ghi) if rule == "contain" else np.all(qmin <= ghi) and np.all(qmax >= glo)
            if ok and len(entries) < max_entries:
                entries.append(ref | (e * 4 + j) << 26)
        e += 1
    return np.array(entries, dtype=np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=200)
    ap.add_argument("--entries", type=int, default=8)
    ap.add_argument("--rule", default="overlap")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--atlas", type=int, default=2048)
    ap.add_argument("--builder", type=int, default=1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc

    flat = FlatScene(scenes.terrain_scene(grid=args.grid, atlas=args.atlas, seed=3))
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.accumulate, opt.normal_fp16, opt.seed = args.spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, 3
    tris = orc.scene_triangles(flat.desc)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    assert len(surfels) == args.atlas * args.atlas
    nodes2 = orc.bvh_build(tris, args.builder, 1)[0] if args.builder != 2 else orc.bvh_build_sah(tris)[0]
    n4, lt = orc.bvh4_build(nodes2, pairs=True)
    _, frame = orc.bvh2q_build(nodes2)
    n4q = orc.bvh4q_build(n4, frame)
    lib = orc.load()
    lib.orc_entry_walk.restype = C.c_int
    lib.orc_entry_walk.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.POINTER(OrcEntryStats)]
    n4q_c, frame_c, lt_c = np.ascontiguousarray(n4q, dtype=np.uint32), np.ascontiguousarray(frame, dtype=np.float32), np.ascontiguousarray(lt)

    rng = np.random.default_rng(7)
    tw = args.atlas // 8
    plain, plain_hit = orc.OrcTraceStats(), orc.OrcTraceStats()
    es = OrcEntryStats()
    lengths = []
    grid = surfels.reshape(args.atlas, args.atlas)
    for t in rng.choice(tw * tw, size=args.tiles, replace=False):
        ty, tx = divmod(int(t), tw)
        sub = np.ascontiguousarray(grid[8 * ty:8 * ty + 8, 8 * tx:8 * tx + 8].reshape(-1))
        pos = sub["position"].astype(np.float64)
        ent = entry_list(n4q, frame, pos.min(axis=0) - abs(opt.ray_bias), pos.max(axis=0) + abs(opt.ray_bias), args.entries, args.rule)
        lengths.append(len(ent))
        rays = np.ascontiguousarray(orc.generate_rays(sub, None, opt, args.atlas))
        a, h = orc.walk4q(n4q, frame, lt, rays, True)
        for s, d in ((a, plain), (h, plain_hit)):
            d.rays += s.rays; d.inner_visits += s.inner_visits; d.leaf_visits += s.leaf_visits; d.hits += s.hits
        rc = lib.orc_entry_walk(n4q_c.ctypes.data, len(n4q_c), frame_c.ctypes.data, lt_c.ctypes.data, rays.ctypes.data, len(rays),
                                ent.ctypes.data, len(ent), C.byref(es))
        assert rc == 0
    assert es.hits == plain.hits and es.rays == plain.rays, (es.hits, plain.hits)
    r = float(plain.rays)
    rec = {"workload": f"BASELINE configs[2]: {len(tris)} triangles, {args.atlas}^2 atlas, {args.spp} spp, any hit; all rays of {args.tiles} random 8x8-texel tiles",
           "builder": args.builder, "rule": args.rule, "max_entries": args.entries, "rays": int(r), "hit_rate": round(plain.hits / r, 4),
           "entries_per_tile_mean": round(float(np.mean(lengths)), 2), "entries_per_tile_histogram": np.bincount(lengths, minlength=9).tolist(),
           "plain_walk": {"inner_visits_per_ray": round(plain.inner_visits / r, 3), "triangle_tests_per_ray": round(plain.leaf_visits / r, 3)},
           "entry_walk": {"entries_tested_per_ray": round(es.entries_tested / r, 3), "mask_bits_per_ray": round(es.mask_bits / r, 3),
                          "inner_visits_per_ray_after_entry_phase": round(es.inner_visits / r, 3), "triangle_tests_per_ray": round(es.leaf_visits / r, 3)},
           "inner_visits_moved_to_entry_phase": round(1.0 - es.inner_visits / plain.inner_visits, 4),
           "same_hits_as_plain_walk": True}
    text = json.dumps(rec, indent=1)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
