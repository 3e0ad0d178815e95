#!/usr/bin/env python3
"""tools/measure_region_cut.py (a measurement script: the oracle as a counter; not a test, not product code) — what a per-tile
REGION CUT would take off the walk of SHORT rays (BASELINE configs[4]: max ray distance = falloff distance 1.25), counted by the
CPU oracle before anything is built for the GPU.

All rays of an 8x8-texel tile start within `bias` of the tile's surfels and are at most tmax long: they stay inside the box R of
the surfels' positions grown by tmax + bias.  Walk the tree with R once per tile: a child whose box does not meet R cannot be
entered by any of the tile's rays and is dropped, the others replace their parent, until the list would outgrow `entries`.
Every ray of the tile then starts with that list on its stack instead of the root.  Same hits, same distances (checked).

  python tools/measure_region_cut.py [--instances 100] [--rock-n 64] [--atlas 2048] [--tiles 300] [--entries 8] [--out FILE]
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


def region_cut(n4q, frame, lo, hi, max_entries):
    origin, scale = frame[:3].astype(np.float64), frame[3:].astype(np.float64)
    cut, final = [0], []
    while cut:
        # expand the entry whose turn it is (breadth first); children that do not meet R are dropped
        node = cut.pop(0)
        w = n4q[node].reshape(4, 4)
        kids = []
        for j in range(4):
            q = w[j, :3]
            qmin, qmax = (q & 0x7fff).astype(np.float64), ((q >> 16) & 0x7fff).astype(np.float64)
            if (w[j, :3] == 0x8000ffff).all():
                continue                                    # empty slot
            bmin, bmax = origin + qmin * scale, origin + qmax * scale
            if np.all(bmin <= hi) and np.all(bmax >= lo):
                kids.append(int(w[j, 3]))
        if len(cut) + len(final) + len(kids) > max_entries:
            final.append(node)                              # no room for its children: the node itself stays an entry
            continue
        for r in kids:
            (final if r >= 0x80000000 else cut).append(r)
    return np.array(final, dtype=np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, default=100)
    ap.add_argument("--rock-n", type=int, default=64)
    ap.add_argument("--atlas", type=int, default=2048)
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--tiles", type=int, default=300)
    ap.add_argument("--entries", type=int, default=8)
    ap.add_argument("--tmax", type=float, default=1.25)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc
    lib = orc.load()
    lib.orc_walk4q_from.restype = C.c_int
    lib.orc_walk4q_from.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]

    t0 = time.time()
    flat = FlatScene(scenes.instanced_scene(instances=a.instances, rock_n=a.rock_n, atlas=a.atlas, seed=5, bpc=4))
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.accumulate, opt.normal_fp16, opt.seed = a.spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, 5
    opt.use_any_hit, opt.falloff_distance, opt.clamp_tmax_to_falloff = 0, a.tmax, 1
    tris = orc.scene_triangles(flat.desc)
    nodes2, depth = orc.bvh_build(tris)
    n4, leaf_tris = orc.bvh4_build(nodes2)
    _, frame = orc.bvh2q_build(nodes2)
    n4q = orc.bvh4q_build(n4, frame)
    leaf_tris = np.ascontiguousarray(np.concatenate([leaf_tris, np.zeros(1, dtype=leaf_tris.dtype)]))
    surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    print(f"{len(tris)} triangles, depth {depth}, {len(n4q)} 4-ary nodes, {len(surfels)} surfels ({time.time() - t0:.0f} s)", file=sys.stderr)
    starts = np.flatnonzero(sidx == 0).tolist() + [len(surfels)]
    section = np.zeros(len(surfels), dtype=np.int64)
    for k, (sb, se) in enumerate(zip(starts[:-1], starts[1:])):
        section[sb:se] = k
    key = (section << 40) | ((surfels["texel_y"].astype(np.int64) // 8) << 20) | (surfels["texel_x"].astype(np.int64) // 8)
    order = np.argsort(key, kind="stable")
    keys, first = np.unique(key[order], return_index=True)
    rng = np.random.default_rng(1)
    pick = rng.choice(len(keys), size=min(a.tiles, len(keys)), replace=False)
    bounds = np.append(first, len(order))

    plain, cut = orc.OrcTraceStats() if hasattr(orc, "OrcTraceStats") else None, None
    class Stats(C.Structure):
        _fields_ = [(n, C.c_uint64) for n in ("rays", "inner_visits", "leaf_visits", "hits")]
    plain, cut = Stats(), Stats()
    tp, tc = C.c_double(0.0), C.c_double(0.0)
    entries_total, inner_entries, dropped_leaf_only = 0, 0, 0
    n4q_c = np.ascontiguousarray(n4q)
    frame_c = np.ascontiguousarray(frame)
    for t in pick:
        idx = order[bounds[t]:bounds[t + 1]]
        sub = np.ascontiguousarray(surfels[idx])
        rays = orc.generate_rays(sub, None, opt, a.atlas, t_max=a.tmax)
        pos = np.stack([sub["position"][:, k] for k in range(3)], axis=1).astype(np.float64) if "position" in sub.dtype.names else None
        if pos is None:
            pos = rays["o"][:, :3].astype(np.float64)
        grow = a.tmax + float(opt.ray_bias) * 2 + 1e-4
        lo, hi = pos.min(0) - grow, pos.max(0) + grow
        lst = region_cut(n4q, frame.astype(np.float64), lo, hi, a.entries)
        entries_total += len(lst); inner_entries += int((lst < 0x80000000).sum())
        root = np.zeros(1, dtype=np.uint32)
        rc = lib.orc_walk4q_from(n4q_c.ctypes.data, len(n4q_c), frame_c.ctypes.data, leaf_tris.ctypes.data, rays.ctypes.data, len(rays), 0,
                                 root.ctypes.data, 1, C.byref(plain), C.byref(tp))
        assert rc == 0
        if len(lst):
            rc = lib.orc_walk4q_from(n4q_c.ctypes.data, len(n4q_c), frame_c.ctypes.data, leaf_tris.ctypes.data, rays.ctypes.data, len(rays), 0,
                                     lst.ctypes.data, len(lst), C.byref(cut), C.byref(tc))
            assert rc == 0
        else:
            cut.rays += int((rays["extra"][:, 1] != 0).sum())
    assert plain.rays == cut.rays and plain.hits == cut.hits and tp.value == tc.value, (plain.hits, cut.hits, tp.value, tc.value)
    rec = {"scene": f"{a.instances} rocks x {2 * a.rock_n * a.rock_n} triangles + ground, {a.atlas}^2 atlas, closest hit, tmax = falloff = {a.tmax}",
           "triangles": len(tris), "bvh_depth": int(depth), "tiles": len(pick), "rays": int(plain.rays), "max_entries": a.entries,
           "plain_walk": {"inner_visits_per_ray": round(plain.inner_visits / plain.rays, 3), "triangle_tests_per_ray": round(plain.leaf_visits / plain.rays, 3)},
           "from_region_cut": {"inner_visits_per_ray": round(cut.inner_visits / cut.rays, 3), "triangle_tests_per_ray": round(cut.leaf_visits / cut.rays, 3),
                               "entries_per_tile": round(entries_total / len(pick), 2), "inner_entries_per_tile": round(inner_entries / len(pick), 2)},
           "inner_visits_ratio": round(cut.inner_visits / max(1, plain.inner_visits), 4), "same_hits_and_distances": True}
    print(json.dumps(rec, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rec, f, indent=1); f.write("\n")


if __name__ == "__main__":
    main()
