#!/usr/bin/env python3
"""tools/measure_horizon_proto2.py (a measurement script, not a test, not product code) — numpy prototype of the THREE-PHASE horizon-table build of k_horizon:

  A  lane = item, world frame: a best-first walk of the tree from the tile's box; boxes nearer than r_near are opened down to their triangles (the NEAR
     list); every other box raises a per-tile world horizon H[az] (sin of the elevation below which a direction may be blocked for SOME origin of the
     tile: the box inflated by the tile's extents) when it is a leaf or small, is dropped when it lies wholly under H, is opened otherwise;
  B  lane = surfel: the near triangles against each surfel's own tangent frame -> its local table (first blocked column per azimuth sector);
  C  lane = surfel: the boundary of {sinE <= H[az]} as a polyline of great-circle arcs, each arc into the local table.

Reports the share of rays culled, the work per tile and that no culled ray is occluded for the oracle.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import measure_horizon_proto as hp

MID = 3.0
VM = 0
GROUPS = 1
ORDER = 'best'
KW = 32            # world azimuth sectors
MU = 2e-3


def az_range_rect(a0, a1, g0, g1):
    """angular hull (lo, hi in turns, hi - lo < 0.5) of a rectangle that does not contain the origin, or None (contains it)"""
    if a0 <= 0 <= a1 and g0 <= 0 <= g1:
        return None
    cs = np.array([[a0, g0], [a0, g1], [a1, g0], [a1, g1]])
    th = np.arctan2(cs[:, 1], cs[:, 0]) / (2 * np.pi)
    rel = (th - th[0] + 0.5) % 1.0 - 0.5
    return th[0] + rel.min(), th[0] + rel.max()


def phase_a(n4, lt, tlo, thi, r_near, thr, stats, Mf=None, flo=None, fhi=None, origins=None):
    """Mf: rows = the tile frame's axes (X, N, Z) in world coordinates; flo, fhi: the surfels' extent in that frame"""
    if Mf is None:
        Mf = np.eye(3); flo, fhi = tlo, thi
    aM = np.abs(Mf)
    fc, te = 0.5 * (flo + fhi), 0.5 * (fhi - flo)
    H = np.full(KW, -1.0)
    near = []
    openl = [(2.0, 0)]
    def sectors(lo, hi):
        a, b = int(np.floor((lo - 1e-3) * KW)), int(np.floor((hi + 1e-3) * KW))
        if b - a + 1 >= KW:
            return list(range(KW))
        return [k % KW for k in range(a, b + 1)]
    while openl:
        if ORDER == 'best':
            openl.sort(key=lambda x: -x[0])
            take, openl = openl[:16], openl[16:]
        else:
            take, openl = openl[-16:], openl[:-16]
        new_children = []
        stats["rounds"] += 1
        for _, node in take:
            nd = n4[node]
            for j in range(4):
                ref = int(nd["lo"][j][3].view(np.uint32))
                if ref == 0xffffffff:
                    continue
                lo, hi = nd["lo"][j][:3].astype(np.float64), nd["hi"][j][:3].astype(np.float64)
                if (lo > hi).any():
                    continue
                stats["box_tests"] += 1
                gap = np.sqrt((np.maximum(np.maximum(lo - thi, tlo - hi), 0) ** 2).sum())
                leaf = bool(ref & 0x80000000)
                if gap <= r_near:
                    if leaf:
                        p = ref & 0x3fffffff
                        near.extend([p, p + 1] if ref & 0x40000000 else [p])
                    else:
                        new_children.append((3.0, ref))
                    continue
                v, e = Mf @ (0.5 * (lo + hi)) - fc, aM @ (0.5 * (hi - lo)) + te
                a0, a1, b1, g0, g1 = v[0] - e[0], v[0] + e[0], v[1] + e[1], v[2] - e[2], v[2] + e[2]
                rx, rz = max(max(a0, -a1), 0.0), max(max(g0, -g1), 0.0)
                rmin2 = rx * rx + rz * rz
                rmax2 = max(a0 * a0, a1 * a1) + max(g0 * g0, g1 * g1)
                r2 = rmin2 if b1 > 0 else rmax2
                top = b1 / np.sqrt(b1 * b1 + r2) if (b1 != 0 or r2 > 0) else 1.0
                top = min(1.0, top + 1e-3)
                rng_ = az_range_rect(a0, a1, g0, g1)
                secs = list(range(KW)) if rng_ is None else sectors(*rng_)
                secs_box = secs
                if all(top <= H[k] for k in secs):
                    stats["dropped"] += 1
                    continue
                ang2 = (e * e).sum() / max((v * v).sum(), 1e-300)
                if leaf and origins is not None and gap <= MID * r_near:
                    # the leaf's triangles from every real origin of the tile (lane = surfel, then a wave reduction)
                    p = ref & 0x3fffffff
                    vs = []
                    for q in range(2 if ref & 0x40000000 else 1):
                        t = lt[p + q]
                        vs += [t["v0"][:3], t["v1"][:3], t["v2"][:3]]
                    vs = np.array(vs, dtype=np.float64) @ Mf.T                       # [nv, 3] in the tile frame
                    if VM:
                        # vertex against the tile's extent in its own frame: nearest point of the rectangle, lowest origin
                        da = np.maximum(np.maximum(flo[0] - vs[:, 0], vs[:, 0] - fhi[0]), 0)
                        dg = np.maximum(np.maximum(flo[2] - vs[:, 2], vs[:, 2] - fhi[2]), 0)
                        rho2 = da * da + dg * dg
                        # the highest the vertex can stand over an origin, the lowest it can (for a vertex below every origin the FARTHEST point counts)
                        db_hi = vs[:, 1] - flo[1]
                        fa = np.maximum(np.abs(vs[:, 0] - flo[0]), np.abs(vs[:, 0] - fhi[0])); fg = np.maximum(np.abs(vs[:, 2] - flo[2]), np.abs(vs[:, 2] - fhi[2]))
                        rho2f = fa * fa + fg * fg
                        el = np.where(db_hi > 0, db_hi / np.sqrt(db_hi * db_hi + rho2 + 1e-300), (vs[:, 1] - fhi[1]) / np.sqrt((vs[:, 1] - fhi[1]) ** 2 + rho2f + 1e-300))
                        edge = max(np.linalg.norm(vs[i] - vs[j]) for i in range(len(vs)) for j in range(i))
                        om = edge / max(np.sqrt(rho2.min()), 1e-9)
                        top = min(1.0, float(el.max()) + om * om / 7.0 + 1e-3)
                        # azimuths: the vertices against the rectangle's four corners
                        cs = np.array([[flo[0], flo[2]], [flo[0], fhi[2]], [fhi[0], flo[2]], [fhi[0], fhi[2]]])
                        dd = vs[:, None, [0, 2]] - cs[None, :, :]
                        th = np.arctan2(dd[:, :, 1], dd[:, :, 0]).reshape(-1) / (2 * np.pi)
                        rel = (th - th[0] + 0.5) % 1.0 - 0.5
                        inside = ((vs[:, 0] >= flo[0]) & (vs[:, 0] <= fhi[0]) & (vs[:, 2] >= flo[2]) & (vs[:, 2] <= fhi[2])).any()
                        if inside or rel.max() - rel.min() >= 0.5:
                            secs = list(range(KW))
                        else:
                            secs = sectors(th[0] + rel.min(), th[0] + rel.max())
                    elif GROUPS:
                        g = origins.reshape(16, 4, 3)
                        glo, ghi = g.min(1), g.max(1)                                  # [16, 3]
                        dlo = vs[None, :, :] - ghi[:, None, :]; dhi = vs[None, :, :] - glo[:, None, :]     # the vertex relative to the group's box: [lo, hi] per axis
                        da = np.maximum(np.maximum(dlo[:, :, 0], -dhi[:, :, 0]), 0); dg = np.maximum(np.maximum(dlo[:, :, 2], -dhi[:, :, 2]), 0)
                        rho2 = da * da + dg * dg
                        fa = np.maximum(np.abs(dlo[:, :, 0]), np.abs(dhi[:, :, 0])); fg = np.maximum(np.abs(dlo[:, :, 2]), np.abs(dhi[:, :, 2]))
                        rho2f = fa * fa + fg * fg
                        bh = dhi[:, :, 1]
                        el = np.where(bh > 0, bh / np.sqrt(bh * bh + rho2 + 1e-300), bh / np.sqrt(bh * bh + rho2f + 1e-300))
                        edge = max(np.linalg.norm(vs[i] - vs[j]) for i in range(len(vs)) for j in range(i))
                        om = edge / max(np.sqrt((rho2 + np.maximum(dlo[:, :, 1], 0) ** 2).min()), 1e-9)
                        top = min(1.0, float(el.max()) + om * om / 7.0 + 1e-3)
                        secs = secs_box
                    else:
                        d = vs[None, :, :] - origins[:, None, :]                          # [64, nv, 3]
                        dist = np.linalg.norm(d, axis=2)
                        el = d[:, :, 1] / dist
                        edge = max(np.linalg.norm(vs[i] - vs[j]) for i in range(len(vs)) for j in range(i))
                        om = edge / dist.min()
                        top = min(1.0, float(el.max()) + om * om / 7.0 + 1e-3)
                        th = np.arctan2(d[:, :, 2], d[:, :, 0]).reshape(-1) / (2 * np.pi)
                        rel = (th - th[0] + 0.5) % 1.0 - 0.5
                        if rel.max() - rel.min() >= 0.5:
                            secs = list(range(KW))
                        else:
                            secs = sectors(th[0] + rel.min(), th[0] + rel.max())
                    stats["applied"] += 1
                    if all(top <= H[k] for k in secs):
                        continue
                    stats["raised"] = stats.get("raised", 0) + 1
                    for k in secs:
                        H[k] = max(H[k], top)
                    continue
                if leaf or ang2 <= thr * thr:
                    stats["applied"] += 1
                    for k in secs:
                        H[k] = max(H[k], top)
                    continue
                new_children.append((top, ref))
        if ORDER == 'lifo':
            new_children.sort(key=lambda x: x[0])
        openl.extend(new_children)
    return H, near


def tile_N(sub):
    n = sub["normal"].astype(np.float64).mean(0)
    return n / np.linalg.norm(n)


def phase_c(tile, H, Mf):
    """the boundary of {sinE <= H[az]} as great-circle arcs -> every surfel's local table"""
    P, Xd, Nd, Zd, K, C = tile.P, tile.Xd, tile.Nd, tile.Zd, tile.K, tile.C
    half = np.pi / KW
    pts = []        # polyline vertices (unit vectors), closed
    for k in range(KW):
        h = H[k]
        E = np.arcsin(np.clip(h, -1, 1))
        if E < 0:
            E = np.arctan(np.tan(E) * np.cos(half))           # the chord of this latitude stays above it
        E = min(np.pi / 2, E + 2e-3)
        for az in (2 * np.pi * k / KW, 2 * np.pi * (k + 1) / KW):
            pts.append(Mf.T @ np.array([np.cos(E) * np.cos(az), np.sin(E), np.cos(E) * np.sin(az)]))
    n = len(pts)
    for i in range(n):
        a, b = pts[i], pts[(i + 1) % n]
        if np.allclose(a, b):
            continue
        # local coordinates of both ends for all 64 surfels (the local frame's dual basis: exact coefficients)
        la = np.stack([Xd @ a, Nd @ a, Zd @ a], 1); lb = np.stack([Xd @ b, Nd @ b, Zd @ b], 1)
        # normalise by the direction's length in local coordinates (N need not be a unit vector)
        # largest cos(theta) = y / |.| along the arc: ends, or the arc's point nearest the local pole
        def cosn(l):
            return l[:, 1] / np.maximum(np.linalg.norm(l, axis=1), 1e-300)
        f = np.maximum(cosn(la), cosn(lb))
        nrm = np.cross(a, b)
        nl = np.linalg.norm(nrm)
        if nl > 1e-12:
            nrm = nrm / nl
            # local pole in world coordinates: the direction d with local coordinates (0, 1, 0) is N itself
            Nw = np.cross(Zd, Xd)                                  # proportional to N (dual of the dual basis), any positive scale
            Nw = Nw / np.linalg.norm(Nw, axis=1, keepdims=True)
            q = Nw - (Nw @ nrm)[:, None] * nrm[None, :]
            ql = np.linalg.norm(q, axis=1)
            qn = q / np.maximum(ql, 1e-300)[:, None]
            # inside the arc: on the same side of both ends
            inside = (np.cross(a, qn) @ nrm >= 0) & (np.cross(qn, b) @ nrm >= 0) & (ql > 1e-9)
            lq = np.stack([(Xd * qn).sum(1), (Nd * qn).sum(1), (Zd * qn).sum(1)], 1)
            f = np.where(inside, np.maximum(f, cosn(lq)), f)
        f = np.clip(f + 1e-3, -1, 1)
        u = np.where(f > 0, 1.0 - f * f, 1.0) - MU
        c0 = np.clip(np.floor(u * C), 0, C).astype(np.int64)
        th_a = np.arctan2(la[:, 2], la[:, 0]) / (2 * np.pi); th_b = np.arctan2(lb[:, 2], lb[:, 0]) / (2 * np.pi)
        rel = (th_b - th_a + 0.5) % 1.0 - 0.5
        lo = np.minimum(th_a, th_a + rel); hi = np.maximum(th_a, th_a + rel)
        full = (np.abs(rel) > 0.49) | (f >= 0.9999)
        M = hp.sector_mask(0.5 * (lo + hi), 0.5 * (hi - lo), full, K)
        tile.apply((f > 0) & (c0 < C), c0, M)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=4)
    ap.add_argument("--atlas", type=int, default=2048)
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--thr", type=float, default=0.08)
    ap.add_argument("--near", type=float, default=2.0, help="near radius in tile diagonals")
    ap.add_argument("--out", default="")
    ap.add_argument("--frame", default="tile")
    ap.add_argument("--exact-leaves", type=int, default=1)
    ap.add_argument("--mid", type=float, default=3.0)
    ap.add_argument("--vm", type=int, default=0)
    ap.add_argument("--groups", type=int, default=1)
    ap.add_argument("--order", default="best")
    args = ap.parse_args()
    from aobaker_amd import capi, scenes
    from aobaker_amd.scene import FlatScene
    from oracle import oracle as orc
    global MID, VM, GROUPS, ORDER
    MID = args.mid
    VM = args.vm
    GROUPS = args.groups
    ORDER = args.order
    atlas = args.atlas
    flat = FlatScene(scenes.terrain_scene(grid=args.grid, atlas=atlas, seed=3))
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.accumulate, opt.seed = args.spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, 3
    tris = orc.scene_triangles(flat.desc)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    nodes2 = orc.bvh_build(tris, 2, 1)[0]
    n4, lt = orc.bvh4_build(nodes2, pairs=True)
    grid = surfels.reshape(atlas, atlas)
    tw = atlas // 8
    rng = np.random.default_rng(11)
    picks = rng.choice(tw * tw, size=args.tiles, replace=False)
    sx = int(np.floor(np.sqrt(args.spp))); sy = args.spp // sx
    K, C = sy, sx
    tot = dict(rays=0, occluded=0, culled=0, culled_near_only=0, wrong=0, rounds=0, box_tests=0, dropped=0, applied=0, raised=0, near_tris=0, survivors=0, passes_lane=0)
    for t_ in picks:
        sub = np.ascontiguousarray(grid[8 * (t_ // tw):8 * (t_ // tw) + 8, 8 * (t_ % tw):8 * (t_ % tw) + 8].reshape(-1))
        rays = np.ascontiguousarray(orc.generate_rays(sub, None, opt, atlas))
        occ = (orc.occluded_rays(nodes2, rays).reshape(64, args.spp) == 1)[:, :sx * sy].reshape(64, sy, sx)
        tile = hp.Tile(sub, K, C, float(opt.ray_bias), np.inf)
        tlo, thi = tile.P.min(0), tile.P.max(0)
        diag = np.linalg.norm(thi - tlo)
        st = dict(rounds=0, box_tests=0, dropped=0, applied=0)
        if args.frame == "tile":
            Nm = tile_N(sub)
            Xm = np.array([-Nm[2], 0.0, Nm[0]]); Xm /= np.linalg.norm(Xm)
            Zm = np.cross(Xm, Nm)
            Mf = np.stack([Xm, Nm, Zm])
        else:
            Mf = np.eye(3)
        fp = tile.P @ Mf.T
        H, near = phase_a(n4, lt, tlo, thi, args.near * diag, args.thr, st, Mf, fp.min(0), fp.max(0), fp if args.exact_leaves else None)
        for p in near:
            t = lt[p]
            tile.tri(np.stack([t["v0"][:3], t["v1"][:3], t["v2"][:3]]).astype(np.float64), "cheap")
        def count(tile_):
            bl = tile_.blocked[:, :C]
            bits = ((bl[:, :, None] >> np.arange(K, dtype=np.uint64)[None, None, :]) & np.uint64(1)).astype(bool)
            free = np.where(bits.any(1), bits.argmax(1), C)
            return free, np.arange(sx)[None, None, :] < free[:, :, None]
        _, c_near = count(tile)
        tot["culled_near_only"] += int(c_near.sum())
        phase_c(tile, H, Mf)
        free, culled = count(tile)
        tot["rays"] += culled.size; tot["culled"] += int(culled.sum()); tot["occluded"] += int(occ.sum()); tot["wrong"] += int((culled & occ).sum())
        for k, v in st.items():
            tot[k] += v
        tot["near_tris"] += len(near)
        rows = sy // 4
        for p_ in range(4):
            n_s = (sx - free[:, p_ * rows:(p_ + 1) * rows]).sum(1)
            tot["survivors"] += int(n_s.sum()); tot["passes_lane"] += int(n_s.max())
    r = float(tot["rays"]); T = float(args.tiles)
    rec = {"workload": f"terrain, {atlas}^2, {args.spp} spp, {args.tiles} tiles; local {K} x {C}, world horizon of {KW} sectors; small = angular radius {args.thr}; near = {args.near} tile diagonals",
           "occluded_share": round(tot["occluded"] / r, 4), "culled_share": round(tot["culled"] / r, 4), "culled_share_near_only": round(tot["culled_near_only"] / r, 4),
           "culled_and_occluded": tot["wrong"],
           "per_tile": {k: round(tot[k] / T, 1) for k in ("rounds", "box_tests", "dropped", "applied", "raised", "near_tris")},
           "generation_occupancy_per_lane_lists": round(tot["survivors"] / (64.0 * max(1, tot["passes_lane"])), 3)}
    text = json.dumps(rec, indent=1)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
