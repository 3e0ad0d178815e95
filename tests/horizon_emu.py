#!/usr/bin/env python3
"""tests/horizon_emu.py — the horizon kernel (aobaker_amd/csrc/horizon_kernel.h) on the CPU (tests/horizon_emu/horizon_emu.cpp: a host thread per lane) against the
oracle's rays: tables of whole tiles of a scene, the share of samples they skip, and the count of skipped cells that hold an occluded ray (must be 0).
Test infrastructure (tests/test_horizon_emu.py imports it); also the tool behind profiles/r06/horizon/*.json.
  python tests/horizon_emu.py [--scene terrain|terrain_small|cube|inward|sphere|soup|rocks] [--tiles 8] [--spp 256]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_LIB = None


def load():
    global _LIB
    if _LIB is None:
        so = os.path.join(ROOT, "tests", "horizon_emu", "libhorizon_emu.so")
        src = [os.path.join(ROOT, "tests", "horizon_emu", "horizon_emu.cpp"), os.path.join(ROOT, "aobaker_amd", "csrc", "horizon_kernel.h"),
               os.path.join(ROOT, "tests", "horizon_emu", "hip", "hip_runtime.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
            subprocess.run(["g++", "-O2", "-std=c++20", "-fPIC", "-shared", "-pthread", "-DAOB_HORIZON_EMU", "-I" + os.path.join(ROOT, "tests", "horizon_emu"),
                            "-I" + os.path.join(ROOT, "aobaker_amd", "csrc"), src[0], "-o", so], check=True)
        _LIB = C.CDLL(so)
        _LIB.emu_horizon.restype = C.c_int
        _LIB.emu_horizon.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    return _LIB


def frames_of(surfels):
    """k_surfel_frames (raster.hip) in numpy float32: [n, 16] floats"""
    n = len(surfels)
    P = surfels["position"].astype(np.float32); N = surfels["normal"].astype(np.float32)
    X = np.tile(np.array([1, 0, 0], np.float32), (n, 1)); Z = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    m = (N[:, 1] < np.float32(0.9999)) & (N[:, 1] > np.float32(-0.9999))
    with np.errstate(all="ignore"):
        x = np.stack([-N[:, 2], np.zeros(n, np.float32), N[:, 0]], 1)
        x = x / np.sqrt((x * x).sum(1, dtype=np.float32))[:, None]
        z = np.cross(x, N).astype(np.float32)
        z = z / np.sqrt((z * z).sum(1, dtype=np.float32))[:, None]
    X[m], Z[m] = x[m], z[m]
    ok = np.isfinite(P).all(1) & np.isfinite(N).all(1) & np.isfinite(X).all(1) & np.isfinite(Z).all(1)
    f = np.zeros((n, 16), np.float32)
    f[:, 0:3] = P; f[:, 3:6] = N; f[:, 6:9] = X; f[:, 9:12] = Z
    f[:, 14] = np.where(ok, np.uint32(1), np.uint32(0)).astype(np.uint32).view(np.float32)
    return f


def tables(surfels, n4q, leaf_tris, qframe, spp, bias=1e-3, reach=np.inf, near=0.5, mid=3.0, small=0.14, big=0.36, counters=None, debug=None):
    """uint8 [n, stride]: the free columns of every azimuth row, surfels taken 64 at a time in the order given"""
    sx = int(np.floor(np.sqrt(spp))); sy = spp // sx
    assert sx * sy == spp and 4 <= sx <= 32 and 4 <= sy <= 32
    f = np.ascontiguousarray(frames_of(surfels))
    stride = 16 if sy <= 16 else 32
    out = np.zeros((len(surfels), stride), np.uint8)
    n4q = np.ascontiguousarray(n4q, dtype=np.uint32); lt = np.ascontiguousarray(leaf_tris); qf = np.ascontiguousarray(qframe, dtype=np.float32)
    cnt = np.zeros(8, np.uint64) if counters is None else counters
    tiles = (len(surfels) + 63) // 64
    load().emu_horizon(f.ctypes.data, 0, len(surfels), n4q.ctypes.data, lt.ctypes.data, qf.ctypes.data, len(n4q), sy, sx, bias, reach, near, mid, small * small, big,
                       out.ctypes.data, stride, cnt.ctypes.data, 0, tiles,
                       None if debug is None else debug["H"].ctypes.data, None if debug is None else debug["tables"].ctypes.data)
    return out


def tile_order(surfels_grid):
    """[H, W] surfels -> the 8x8-texel tiles one after the other"""
    H, W = surfels_grid.shape
    return np.ascontiguousarray(surfels_grid.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1))


def scene_of(name, atlas):
    from aobaker_amd import scenes
    if name == "terrain":
        return scenes.terrain_scene(grid=708, atlas=atlas, seed=3)
    if name == "terrain_small":
        return scenes.terrain_scene(grid=64, atlas=atlas, seed=3)
    if name == "cube":
        return scenes.cube_scene(atlas)
    if name == "inward":
        return scenes.cube_scene(atlas, inward=True)
    if name == "sphere":
        return scenes.sphere_scene(n=40, atlas=atlas)
    if name == "soup":
        return scenes.soup_scene(atlas=atlas)
    if name == "rocks":
        return scenes.instanced_scene(instances=9, rock_n=24, atlas=atlas)
    if name == "floor_wall":
        return scenes.floor_wall_scene(atlas=atlas)
    raise SystemExit("unknown scene " + name)


def check(scene, spp, tiles, seed=1, builder=1, verbose=False, **kw):
    """bakes nothing: tables of `tiles` random 64-surfel runs of the scene's first material against the oracle's rays"""
    from aobaker_amd import capi
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc
    flat = FlatScene(scene)
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.accumulate, opt.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, 3
    tris = orc.scene_triangles(flat.desc)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    nodes2 = orc.bvh_build(tris, builder, 1)[0]
    n4, lt = orc.bvh4_build(nodes2, pairs=True)
    _, qframe = orc.bvh2q_build(nodes2)
    n4q = orc.bvh4q_build(n4, qframe)
    W, H = flat.material_array[0].width, flat.material_array[0].height
    if len(surfels) == W * H and W % 8 == 0 and H % 8 == 0:
        surfels = tile_order(surfels.reshape(H, W))
    ntiles = (len(surfels) + 63) // 64
    rng = np.random.default_rng(seed)
    picks = np.sort(rng.choice(ntiles, size=min(tiles, ntiles), replace=False))
    sub = np.concatenate([surfels[64 * t:64 * t + 64] for t in picks])
    cnt = np.zeros(8, np.uint64)
    tab = tables(sub, n4q, lt, qframe, spp, bias=float(opt.ray_bias), counters=cnt, **kw)
    sx = int(np.floor(np.sqrt(spp))); sy = spp // sx
    rays = np.ascontiguousarray(orc.generate_rays(sub, None, opt, W))
    occ = (orc.occluded_rays(nodes2, rays).reshape(len(sub), spp) == 1).reshape(len(sub), sy, sx)
    free = tab[:, :sy].astype(np.int64)
    skipped = np.arange(sx)[None, None, :] < free[:, :, None]
    T = float(max(1, int(cnt[5])))
    rec = {"surfels": int(len(sub)), "rays": int(skipped.size), "skipped_share": round(float(skipped.mean()), 4), "occluded_share": round(float(occ.mean()), 4),
           "skipped_cells_with_an_occluded_ray": int((skipped & occ).sum()),
           "per_tile": {"rounds": round(int(cnt[0]) / T, 1), "box_tests": round(int(cnt[1]) / T, 1), "near_triangles": round(int(cnt[2]) / T, 1),
                        "middle_leaves": round(int(cnt[3]) / T, 1), "given_up": round(int(cnt[4]) / T, 3)}}
    if verbose and rec["skipped_cells_with_an_occluded_ray"]:
        bad = np.argwhere(skipped & occ)
        rec["first_bad"] = [[int(v) for v in b] for b in bad[:8]]
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="terrain")
    ap.add_argument("--atlas", type=int, default=0)
    ap.add_argument("--tiles", type=int, default=8)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--near", type=float, default=0.5)
    ap.add_argument("--mid", type=float, default=3.0)
    ap.add_argument("--builder", type=int, default=1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    atlas = args.atlas or {"terrain": 2048, "terrain_small": 256}.get(args.scene, 64)
    rec = check(scene_of(args.scene, atlas), args.spp, args.tiles, seed=args.seed, builder=args.builder, verbose=True, near=args.near, mid=args.mid)
    rec = {"scene": args.scene, "atlas": atlas, "spp": args.spp, **rec}
    text = json.dumps(rec, indent=1)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
