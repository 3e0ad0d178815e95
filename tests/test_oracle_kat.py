"""Known-answer tests that pin the CPU oracle.

The reference has no tests or golden vectors of its own, so the oracle is pinned by
(1) the published MSVC rand() sequence its sampler draws from, (2) the arithmetic of the in-tree
OpenCL primitives it restates, checked on their edge cases, and (3) closed-form geometry.
"""
import math

import numpy as np
import pytest

from aobaker_amd import capi, scenes
from aobaker_amd.scene import FlatScene


def opts(orc, **kw):
    o = orc.default_options()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


# ---- (9) MSVC rand ---------------------------------------------------------------
def test_msvc_rand_sequence(orc):
    # the first values of the unseeded MSVC CRT rand(), which BakeEngine.cpp:316-317 consumes
    assert orc.msvc_rand_sequence(8) == [41, 18467, 6334, 26500, 19169, 15724, 11478, 29358]


def test_hemisphere_table_matches_formula(orc):
    # BakeEngine.cpp:312-333: x = s*cos(phi), z = s*sin(phi), y = 1 - s (NOT cos(theta))
    t = orc.hemisphere_table(1000)
    seq = orc.msvc_rand_sequence(2000)
    u = np.float32(seq[0]) / np.float32(32767.0)
    v = np.float32(seq[1]) / np.float32(32767.0)
    s = np.sqrt(np.float32(-u * (u - np.float32(2.0))))
    phi = np.float32(2.0 * 3.14159265358979323846) * v
    assert abs(t[0, 0] - s * math.cos(phi)) < 1e-6 and abs(t[0, 2] - s * math.sin(phi)) < 1e-6
    r = np.sqrt(t[:, 0] ** 2 + t[:, 2] ** 2)
    assert np.allclose(t[:, 1], 1.0 - r, atol=1e-6)
    assert (t[:, 1] >= -1e-6).all() and (t[:, 1] <= 1.0).all()
    # |dir|^2 = s^2 + (1-s)^2 lies in [0.5, 1]: the directions are deliberately unnormalised
    n2 = (t ** 2).sum(axis=1)
    assert n2.min() >= 0.5 - 1e-5 and n2.max() <= 1.0 + 1e-5


def test_compat_table_size(orc):
    # BakeEngine.cpp:359-364: chunkTexelMax = min(rayChunkSize / sampleCount, 65535)
    assert orc.load().orc_compat_table_entries(opts(orc, sample_count=100)) == 65535 * 100
    assert orc.load().orc_compat_table_entries(opts(orc, sample_count=1024)) == (0xFFFFFF // 1024) * 1024


# ---- D3D ftoi and fp16 -------------------------------------------------------------
def test_ftoi(orc):
    assert orc.ftoi(254.999) == 254 and orc.ftoi(-3.9) == -3 and orc.ftoi(0.0) == 0
    assert orc.ftoi(float("nan")) == 0
    assert orc.ftoi(float("inf")) == 2**31 - 1 and orc.ftoi(-float("inf")) == -2**31
    assert orc.ftoi(3e9) == 2**31 - 1


def test_fp16_rne_matches_numpy(orc):
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(2000).astype(np.float32) * np.float32(10) ** rng.integers(-8, 5, 2000),
                           np.float32([0.0, -0.0, 1.0, 0.99999994, 65504.0, 65519.9, 65520.0, 1e-8, 6e-8, 6.1e-5, np.inf, -np.inf])])
    for v in vals:
        want = np.float32(v).astype(np.float16).view(np.uint16)
        assert orc.f32_to_f16(float(v), capi.FP16_RNE) == int(want), v
    for h in list(range(0, 0x7c01, 37)) + [0x8000, 0xfbff, 0x0001, 0x03ff, 0x0400]:
        want = np.uint16(h).view(np.float16).astype(np.float32)
        assert orc.f16_to_f32(h) == float(want)


def test_fp16_rtz_never_rounds_away(orc):
    rng = np.random.default_rng(1)
    for v in rng.standard_normal(2000).astype(np.float32):
        h = orc.f32_to_f16(float(v), capi.FP16_RTZ)
        back = orc.f16_to_f32(h)
        assert abs(back) <= abs(float(v))
        assert abs(float(v)) - abs(back) < abs(float(v)) * 2.0 ** -10 + 1e-8
    assert orc.f32_to_f16(0.99999994, capi.FP16_RTZ) == 0x3bff      # the quirk that motivates the RNE default
    assert orc.f32_to_f16(1e6, capi.FP16_RTZ) == 0x7bff and orc.f32_to_f16(1e6, capi.FP16_RNE) == 0x7c00


# ---- (5) Möller–Trumbore edge cases, common.cl:193-216 ------------------------------------
V1, V2, V3 = (0, 0, 0), (1, 0, 0), (0, 1, 0)


def test_triangle_hit_and_miss(orc):
    assert orc.intersect_triangle((0.25, 0.25, 1), (0, 0, -1), V1, V2, V3, 10.0) == pytest.approx(1.0)
    assert orc.intersect_triangle((0.75, 0.75, 1), (0, 0, -1), V1, V2, V3, 10.0) == 10.0     # b1+b2 > 1
    assert orc.intersect_triangle((0.25, 0.25, -1), (0, 0, -1), V1, V2, V3, 10.0) == 10.0    # behind the origin
    # no back-face culling (doBackfaceCulling = 0, GenerateRaysComputeShader.hlsl:52)
    assert orc.intersect_triangle((0.25, 0.25, -1), (0, 0, 1), V1, V2, V3, 10.0) == pytest.approx(1.0)


def test_triangle_parallel_ray_is_a_miss(orc):
    assert orc.intersect_triangle((0.25, 0.25, 1), (1, 0, 0), V1, V2, V3, 10.0) == 10.0      # denom == 0


def test_triangle_edges_inclusive(orc):
    # b1 == 0, b2 == 0 and b1 + b2 == 1 are all accepted (the comparisons are strict)
    assert orc.intersect_triangle((0.0, 0.5, 1), (0, 0, -1), V1, V2, V3, 10.0) == pytest.approx(1.0)
    assert orc.intersect_triangle((0.5, 0.0, 1), (0, 0, -1), V1, V2, V3, 10.0) == pytest.approx(1.0)
    assert orc.intersect_triangle((0.5, 0.5, 1), (0, 0, -1), V1, V2, V3, 10.0) == pytest.approx(1.0)
    # t == t_max is accepted, t just beyond is not
    assert orc.intersect_triangle((0.25, 0.25, 1), (0, 0, -1), V1, V2, V3, 1.0) == 1.0
    assert orc.intersect_triangle((0.25, 0.25, 1.5), (0, 0, -1), V1, V2, V3, 1.0) == 1.0     # returns t_max = miss
    # t scales with the (unnormalised) direction: the reference never normalises its rays
    assert orc.intersect_triangle((0.25, 0.25, 1), (0, 0, -0.5), V1, V2, V3, 10.0) == pytest.approx(2.0)


def test_degenerate_triangle_never_hit(orc):
    # the reference's importer feeds (0,0,0) triangles (BakeObject.cpp:96,127): denom == 0
    assert orc.intersect_triangle((0, 0, 1), (0, 0, -1), (0, 0, 0), (0, 0, 0), (0, 0, 0), 5.0) == 5.0


# ---- (6) slab test with zero direction components, common.cl:220-246 ----------------------
def test_safe_invdir_clamps(orc):
    inv = orc.safe_invdir((0.0, -0.0, 2.0))
    assert inv[0] == np.float32(1.0) / np.float32(1e-8)
    assert inv[1] == -(np.float32(1.0) / np.float32(1e-8))
    assert inv[2] == 0.5


def test_bbox_axis_parallel_ray(orc):
    d = (0.0, 0.0, 1.0)
    inv = orc.safe_invdir(d)
    for o, expect in (((0.5, 0.5, -1.0), True), ((1.5, 0.5, -1.0), False), ((0.5, 0.5, 2.0), False)):
        ox = -np.float32(o) * inv
        t0, t1 = orc.intersect_bbox((0, 0, 0), (1, 1, 1), inv, ox, 100.0)
        assert (t0 <= t1) == expect
    # t_max clips the far side, 0 clips the near side
    ox = -np.float32((0.5, 0.5, -1.0)) * inv
    t0, t1 = orc.intersect_bbox((0, 0, 0), (1, 1, 1), inv, ox, 0.5)
    assert t0 == pytest.approx(1.0) and t1 == 0.5 and not t0 <= t1
    ox = -np.float32((0.5, 0.5, 0.5)) * inv
    t0, t1 = orc.intersect_bbox((0, 0, 0), (1, 1, 1), inv, ox, 100.0)
    assert t0 == 0.0 and t1 == pytest.approx(0.5)


# ---- COSINE sampler ------------------------------------------------------------------------
def test_sincos_polynomial(orc):
    for u in np.linspace(0, 0.999, 200):
        s, c = orc.sincos2pi(float(np.float32(u)))
        assert abs(s - math.sin(2 * math.pi * np.float32(u))) < 2e-6
        assert abs(c - math.cos(2 * math.pi * np.float32(u))) < 2e-6


def test_cosine_sampler_distribution(orc):
    spp = 1024
    d = np.array([orc.cosine_sample(5, 1234, i, spp) for i in range(spp)])
    assert np.allclose((d ** 2).sum(axis=1), 1.0, atol=1e-5)              # unit vectors
    assert (d[:, 1] >= 0).all()                                             # upper hemisphere
    assert abs(d[:, 1].mean() - 2.0 / 3.0) < 0.01                           # E[cos] under a cosine pdf
    assert abs(d[:, 0].mean()) < 0.02 and abs(d[:, 2].mean()) < 0.02
    # counter-based: depends on (seed, texel, sample) only
    assert np.array_equal(orc.cosine_sample(5, 1234, 17, spp), d[17])
    assert not np.array_equal(orc.cosine_sample(6, 1234, 17, spp), d[17])
    assert not np.array_equal(orc.cosine_sample(5, 1235, 17, spp), d[17])


# ---- (7) rasterisation rules --------------------------------------------------------------
def test_raster_top_left_rule_and_last_triangle_wins(orc):
    from aobaker_amd.scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices
    # two texel-aligned right triangles forming the square [2,6]x[2,6] of an 8x8 atlas, plus a
    # second copy of the first triangle at a different height drawn LAST
    uv = np.float32([[2, 2], [6, 2], [6, 6], [2, 6]]) / 8
    p = np.float32([[0, 0, 0], [1, 0, 0], [1, 0, 1], [0, 0, 1]])
    p2 = p + np.float32([0, 5, 0])
    pos = np.concatenate([p, p2]); nrm = np.tile(np.float32([0, 1, 0]), (8, 1)); uvs = np.concatenate([uv, uv])
    idx = np.uint32([0, 1, 2, 0, 2, 3, 4, 5, 6])
    sc = Scene(); m = sc.AddMaterial(MaterialObject("A", 8, 8))
    o = sc.AddObject(BakeObject("O")); o.AddSection(Section("s", RenderMesh(make_vertices(pos, nrm, uvs), idx), m))
    surfels, sidx = orc.rasterize(FlatScene(sc).desc, 0, opts(orc, normal_fp16=capi.FP16_OFF))
    # exactly the 16 texels whose centres lie inside: every centre is covered once (shared diagonal
    # and shared edges produce neither holes nor doubles), none outside
    assert len(surfels) == 16
    xy = set(zip(surfels["texel_x"].tolist(), surfels["texel_y"].tolist()))
    assert xy == {(x, y) for x in range(2, 6) for y in range(2, 6)}
    # row-major extraction order (BakeEngine.cpp:569-588)
    assert list(zip(surfels["texel_y"].tolist(), surfels["texel_x"].tolist())) == sorted(zip(surfels["texel_y"].tolist(), surfels["texel_x"].tolist()))
    assert np.array_equal(sidx, np.arange(16))
    # the third triangle (same UVs as the first, drawn last) wins where it covers: y = 5 there
    upper = surfels["position"][:, 1] == 5.0
    for s, up in zip(surfels, upper):
        x, y = s["texel_x"] + 0.5, s["texel_y"] + 0.5
        in_first = (x - 2) >= (y - 2)          # triangle (0,1,2): u >= v side of the diagonal
        assert up == in_first or (x - 2) == (y - 2)
    # interpolated world position = the affine map of the texel centre
    s0 = surfels[0]
    assert s0["texel_x"] == 2 and s0["texel_y"] == 2
    assert np.allclose(s0["position"][[0, 2]], [(2.5 - 2) / 4, (2.5 - 2) / 4], atol=1e-6)


def test_raster_edge_on_texel_centre(orc):
    """A vertical edge through texel centres belongs to the triangle on its right (left edge)."""
    from aobaker_amd.scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices
    uv = np.float32([[2.5, 0], [6.5, 0], [6.5, 8], [2.5, 8]]) / 8
    p = np.float32([[0, 0, 0], [1, 0, 0], [1, 0, 1], [0, 0, 1]])
    sc = Scene(); m = sc.AddMaterial(MaterialObject("A", 8, 8))
    o = sc.AddObject(BakeObject("O"))
    o.AddSection(Section("s", RenderMesh(make_vertices(p, np.tile(np.float32([0, 1, 0]), (4, 1)), uv), np.uint32([0, 1, 2, 0, 2, 3])), m))
    surfels, _ = orc.rasterize(FlatScene(sc).desc, 0, opts(orc))
    xs = sorted(set(surfels["texel_x"].tolist()))
    assert xs == [2, 3, 4, 5]        # x=2 (centre 2.5 on the left edge) in, x=6 (centre 6.5 on the right edge) out
    assert len(surfels) == 4 * 8


def _quads_scene(quads, atlas=8):
    """quads: [(x0, y0, x1, y1, height)] in texel units, each two triangles of a horizontal rectangle at world y = height."""
    from aobaker_amd.scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices
    pos, uvs, idx = [], [], []
    for x0, y0, x1, y1, hgt in quads:
        b = len(pos)
        for x, y in ((x0, y0), (x1, y0), (x1, y1), (x0, y1)):
            pos.append([x, hgt, y]); uvs.append([x / atlas, y / atlas])
        idx += [b, b + 1, b + 2, b, b + 2, b + 3]
    sc = Scene(); m = sc.AddMaterial(MaterialObject("A", atlas, atlas))
    o = sc.AddObject(BakeObject("O"))
    o.AddSection(Section("s", RenderMesh(make_vertices(np.float32(pos), np.tile(np.float32([0, 1, 0]), (len(pos), 1)), np.float32(uvs)), np.uint32(idx)), m))
    return FlatScene(sc)


def test_conservative_raster_gives_a_sliver_its_texels(orc):
    """AOB_RASTER_CONSERVATIVE: a chart thinner than a texel covers no texel centre — REFERENCE finds nothing, CONSERVATIVE
    every texel whose square it overlaps, sampled inside the chart."""
    flat = _quads_scene([(1.1, 3.6, 6.9, 3.9, 2.0)])
    ref, _ = orc.rasterize(flat.desc, 0, opts(orc, normal_fp16=capi.FP16_OFF))
    assert len(ref) == 0
    got, sidx = orc.rasterize(flat.desc, 0, opts(orc, normal_fp16=capi.FP16_OFF, raster_mode=capi.RASTER_CONSERVATIVE))
    assert list(zip(got["texel_x"].tolist(), got["texel_y"].tolist())) == [(x, 3) for x in range(1, 7)]
    assert np.array_equal(sidx, np.arange(6))
    # the sample lies in the chart: world (x, z) = (u, v) in texels here
    assert (got["position"][:, 0] >= 1.1 - 1e-5).all() and (got["position"][:, 0] <= 6.9 + 1e-5).all()
    assert (got["position"][:, 2] >= 3.6 - 1e-5).all() and (got["position"][:, 2] <= 3.9 + 1e-5).all()
    assert np.allclose(got["position"][:, 1], 2.0, atol=1e-6)
    # the nearest point of the winning (last drawn) triangle: on its diagonal, all but straight below the texel centre
    assert np.allclose(got["position"][:, 0], got["texel_x"] + 0.5, atol=0.02)


def test_conservative_raster_touching_is_not_overlapping(orc):
    """A chart whose edges lie on texel boundaries gains no ring of extra texels, and the surfels of the texels whose
    centre it covers are the REFERENCE ones, bit for bit."""
    flat = _quads_scene([(2, 2, 6, 6, 0.0)])
    ref, _ = orc.rasterize(flat.desc, 0, opts(orc))
    got, _ = orc.rasterize(flat.desc, 0, opts(orc, raster_mode=capi.RASTER_CONSERVATIVE))
    assert len(ref) == 16 and got.tobytes() == ref.tobytes()
    # half a texel further out on every side: one ring more, 36 texels
    flat = _quads_scene([(1.5, 1.5, 6.5, 6.5, 0.0)])
    ref, _ = orc.rasterize(flat.desc, 0, opts(orc))
    got, _ = orc.rasterize(flat.desc, 0, opts(orc, raster_mode=capi.RASTER_CONSERVATIVE))
    assert len(ref) == 25 and len(got) == 36            # REFERENCE: the centres on the left and top edges are in (top-left rule)
    inner = (got["texel_x"] >= 1) & (got["texel_x"] < 6) & (got["texel_y"] >= 1) & (got["texel_y"] < 6)
    assert got[inner].tobytes() == ref.tobytes()
    # the ring's samples lie on the chart's border, nearest to their texel centres
    ring = got[~inner]
    assert np.allclose(ring["position"][:, [0, 2]], np.clip(np.stack([ring["texel_x"], ring["texel_y"]], 1) + 0.5, 1.5, 6.5), atol=1e-5)


def test_conservative_raster_centre_coverage_wins(orc):
    """A triangle that covers a texel's centre keeps the texel against a LATER triangle that only overlaps its square."""
    flat = _quads_scene([(2, 2, 6, 6, 0.0), (1.0, 4.1, 7.0, 4.3, 9.0)])         # the sliver is drawn last, at height 9
    got, _ = orc.rasterize(flat.desc, 0, opts(orc, normal_fp16=capi.FP16_OFF, raster_mode=capi.RASTER_CONSERVATIVE))
    by_texel = {(int(s["texel_x"]), int(s["texel_y"])): float(s["position"][1]) for s in got}
    assert len(by_texel) == 16 + 2
    assert all(by_texel[(x, y)] == 0.0 for x in range(2, 6) for y in range(2, 6))
    assert by_texel[(1, 4)] == 9.0 and by_texel[(6, 4)] == 9.0
    # among the triangles that only overlap, the last drawn wins (REFERENCE's rule within the class)
    flat = _quads_scene([(1.0, 4.1, 7.0, 4.3, 9.0), (1.0, 4.6, 7.0, 4.8, 5.0)])
    got, _ = orc.rasterize(flat.desc, 0, opts(orc, normal_fp16=capi.FP16_OFF, raster_mode=capi.RASTER_CONSERVATIVE))
    assert len(got) == 6 and np.allclose(got["position"][:, 1], 5.0, atol=1e-5)


def test_all_layers_keeps_every_fragment_of_the_peel(orc):
    """AOB_LAYERS_ALL: layer j holds each texel's j-th covering triangle in index order; every layer is a section of its own
    (surfels numbered from 0, row-major); LAST is the reference's last fragment."""
    # three charts over each other in ONE section: [2,6]^2 at height 0, [2,6]^2 at height 5, [4,8]x[2,6] at height 9
    flat = _quads_scene([(2, 2, 6, 6, 0.0), (2, 2, 6, 6, 5.0), (4, 2, 8, 6, 9.0)])
    last, _ = orc.rasterize(flat.desc, 0, opts(orc, normal_fp16=capi.FP16_OFF))
    assert len(last) == 24 and set(np.unique(last["position"][:, 1]).tolist()) == {5.0, 9.0}
    got, sidx = orc.rasterize(flat.desc, 0, opts(orc, normal_fp16=capi.FP16_OFF, raster_layers=capi.LAYERS_ALL))
    # layer 0: 16 texels at height 0 and the 8 texels only the third chart covers; layer 1: 16 at height 5; layer 2: the 8 shared by all
    assert len(got) == 24 + 16 + 8
    assert sidx.tolist() == list(range(24)) + list(range(16)) + list(range(8))
    l0, l1, l2 = got[:24], got[24:40], got[40:]
    h = lambda a: a["position"][:, 1]
    assert ((h(l0) == 0.0) == (l0["texel_x"] < 6)).all() and ((h(l0) == 9.0) == (l0["texel_x"] >= 6)).all()
    assert (h(l1) == 5.0).all() and (h(l2) == 9.0).all() and (l2["texel_x"] >= 4).all() and (l2["texel_x"] < 6).all()
    for layer in (l0, l1, l2):
        cells = layer["texel_y"].astype(np.int64) * 8 + layer["texel_x"]
        assert np.all(np.diff(cells) > 0)
    # the last layer of every texel is the reference's surfel
    by_cell = {}
    for s in got:
        by_cell[(int(s["texel_x"]), int(s["texel_y"]))] = s.tobytes()
    assert all(by_cell[(int(s["texel_x"]), int(s["texel_y"]))] == s.tobytes() for s in last)
    # without overlaps the two modes agree
    flat = _quads_scene([(2, 2, 6, 6, 0.0)])
    a, _ = orc.rasterize(flat.desc, 0, opts(orc)); b, _ = orc.rasterize(flat.desc, 0, opts(orc, raster_layers=capi.LAYERS_ALL))
    assert a.tobytes() == b.tobytes()


def test_all_layers_average_in_the_bake(orc):
    """Two coincident charts, the lower one under a lid: LAST bakes only the upper (open) one, ALL averages the two."""
    from aobaker_amd.scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices
    def scene():
        # a floor chart at y = 0 under a big lid at y = 0.5, and the same chart again at y = 1 above the lid, drawn last
        pos = np.float32([[0, 0, 0], [1, 0, 0], [1, 0, 1], [0, 0, 1], [0, 1, 0], [1, 1, 0], [1, 1, 1], [0, 1, 1],
                          [-50, .5, -50], [50, .5, -50], [50, .5, 50], [-50, .5, 50]])
        uv = np.float32([[0, 0], [1, 0], [1, 1], [0, 1]] * 2 + [[2, 2]] * 4)                  # the lid has no texels
        idx = np.uint32([0, 1, 2, 0, 2, 3, 4, 5, 6, 4, 6, 7, 8, 9, 10, 8, 10, 11])
        sc = Scene(); m = sc.AddMaterial(MaterialObject("A", 8, 8, 1, 4))
        o = sc.AddObject(BakeObject("O")); o.AddSection(Section("s", RenderMesh(make_vertices(pos, np.tile(np.float32([0, 1, 0]), (12, 1)), uv), idx), m))
        return sc
    kw = dict(sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, sample_count=16, bleed_distance=0)
    a, b = scene(), scene()
    orc.bake(FlatScene(a).desc, opts(orc, **kw))
    orc.bake(FlatScene(b).desc, opts(orc, raster_layers=capi.LAYERS_ALL, **kw))
    assert np.array_equal(a.GetMaterial(0).GetAOTexture().ao(), np.ones((8, 8), np.float32))     # only the open upper chart
    assert np.array_equal(b.GetMaterial(0).GetAOTexture().ao(), np.full((8, 8), 0.5, np.float32))  # (0 + 1) / 2


def test_conservative_coverage_equals_brute_force(orc):
    """Which texels get a surfel, against an exact restatement in Python integers: open texel square vs triangle by the
    separating-axis test, centre coverage by the top-left rule."""
    from aobaker_amd.scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices
    rng = np.random.default_rng(11)
    n, W = 60, 32
    uvc = rng.uniform(0, 1, (n, 1, 2)); uv = (uvc + rng.normal(scale=0.03, size=(n, 3, 2)) * rng.uniform(0.02, 1.0, (n, 1, 2))).astype(np.float32).reshape(-1, 2)
    pos = np.concatenate([uv, np.zeros((3 * n, 1), np.float32)], axis=1)
    sc = Scene(); m = sc.AddMaterial(MaterialObject("A", W, W))
    o = sc.AddObject(BakeObject("O")); o.AddSection(Section("s", RenderMesh(make_vertices(pos, np.tile(np.float32([0, 0, 1]), (3 * n, 1)), uv), np.arange(3 * n, dtype=np.uint32)), m))
    got, _ = orc.rasterize(FlatScene(sc).desc, 0, opts(orc, raster_mode=capi.RASTER_CONSERVATIVE))
    plain, _ = orc.rasterize(FlatScene(sc).desc, 0, opts(orc))
    snap = lambda v: int(np.rint(np.float32(np.float32(v) * np.float32(W)) * np.float32(256)))
    edge = lambda a, b, p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
    touched, centred = set(), set()
    for t in range(n):
        P = [(snap(uv[3 * t + k][0]), snap(uv[3 * t + k][1])) for k in range(3)]
        area = edge(P[0], P[1], P[2])
        if area == 0:
            continue
        if area < 0:
            P[1], P[2] = P[2], P[1]
        xs, ys = [p[0] for p in P], [p[1] for p in P]
        for py in range(W):
            for px in range(W):
                c = (px * 256 + 128, py * 256 + 128)
                es = [edge(P[k], P[(k + 1) % 3], c) for k in range(3)]
                d = [(P[(k + 1) % 3][0] - P[k][0], P[(k + 1) % 3][1] - P[k][1]) for k in range(3)]
                if all(e >= (0 if (dy < 0 or (dy == 0 and dx > 0)) else 1) for e, (dx, dy) in zip(es, d)):
                    centred.add((py, px))
                box = min(xs) < c[0] + 128 and max(xs) > c[0] - 128 and min(ys) < c[1] + 128 and max(ys) > c[1] - 128
                if box and all(e + 128 * (abs(dx) + abs(dy)) > 0 for e, (dx, dy) in zip(es, d)):
                    touched.add((py, px))
    assert sorted(zip(plain["texel_y"].tolist(), plain["texel_x"].tolist())) == sorted(centred)
    assert list(zip(got["texel_y"].tolist(), got["texel_x"].tolist())) == sorted(touched | centred) and len(touched) > 1.5 * len(centred)


@pytest.mark.parametrize("make", [lambda: scenes.sphere_scene(n=24, atlas=96), lambda: scenes.terrain_scene(grid=40, atlas=64),
                                  lambda: scenes.cube_scene(64)])
def test_conservative_raster_extends_the_reference_coverage(orc, make):
    flat = FlatScene(make())
    ref, _ = orc.rasterize(flat.desc, 0, opts(orc))
    got, _ = orc.rasterize(flat.desc, 0, opts(orc, raster_mode=capi.RASTER_CONSERVATIVE))
    W = flat.material_array[0].width
    cell = lambda a: a["texel_y"].astype(np.int64) * W + a["texel_x"]
    keep = np.isin(cell(got), cell(ref))
    assert keep.sum() == len(ref) and got[keep].tobytes() == ref.tobytes() and len(got) >= len(ref)
    assert np.all(np.diff(cell(got)) > 0)                                       # row-major, one surfel per texel


# ---- (1)(2) trivial scenes -------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["compat", "cosine"])
def test_open_plane_is_white(orc, mode):
    sc = scenes.plane_scene(16, bpc=1)
    o = opts(orc, sample_count=32)
    if mode == "cosine":
        o.sampler, o.accumulate = capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT
    st = orc.bake(FlatScene(sc).desc, o)
    assert st.hits == 0 and st.rays == 16 * 16 * 32
    assert (sc.GetMaterial(0).GetAOTexture().GetData() == 255).all()


def test_closed_cube_is_black_inside(orc):
    sc = scenes.cube_scene(64, inward=True)
    o = opts(orc, sample_count=32, sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT)
    st = orc.bake(FlatScene(sc).desc, o)
    tex = sc.GetMaterial(0).GetAOTexture().as_array()[..., 0]
    assert st.hits >= st.rays * 0.999          # only rays through the (non-watertight) edges escape
    assert tex.max() <= 8                      # covered texels ~0, gutters bleed ~0


# ---- (3) floor under a ceiling, falloff scoring ----------------------------------------------
def test_floor_under_ceiling_falloff_closed_form(orc):
    """Cosine sampler, falloff mode: AO = E[min(t,F)/F], t = h/cos(theta), cos^2 uniform.
    With c0 = h/F:  AO = c0 * 2 * (1 - c0) + c0^2 ... evaluated by the exact integral below."""
    h, F = 0.5, 1.25
    sc = scenes.floor_ceiling_scene(8, height=h, bpc=4)
    o = opts(orc, sample_count=4096, sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, use_any_hit=0,
             falloff_distance=F, normal_fp16=capi.FP16_OFF)
    orc.bake(FlatScene(sc).desc, o, threads=4)
    ao = sc.GetMaterial(0).GetAOTexture().as_array()[..., 0]
    # pdf of c = cos(theta) under cosine weighting is 2c on [0,1]; hit at t = h/c; score t/F if t<F else 1
    c0 = h / F
    expect = c0 * c0 + (2 * h / F) * (1 - c0)           # int_0^c0 2c dc  +  int_c0^1 (h/(cF)) 2c dc
    assert abs(ao.mean() - expect) < 3e-3
    assert np.abs(ao - expect).max() < 0.02


# ---- section visibility and material range -----------------------------------
def _ao(orc, sc, **kw):
    o = opts(orc, sample_count=64, sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, normal_fp16=capi.FP16_OFF, **kw)
    orc.bake(FlatScene(sc).desc, o, threads=4)
    return sc.GetMaterial(0).GetAOTexture().as_array()[..., 0]


def test_section_hidden_from_others_occludes_nobody(orc):
    sc = scenes.floor_ceiling_scene(8, height=0.5, bpc=4)
    assert (_ao(orc, sc) == 0.0).all()                                    # any hit: every ray meets the ceiling
    sc = scenes.floor_ceiling_scene(8, height=0.5, bpc=4)
    sc.GetObject(1).GetSection(0).SetVisibleToOthers(False)               # the ceiling
    assert (_ao(orc, sc) == 1.0).all()
    sc = scenes.floor_ceiling_scene(8, height=0.5, bpc=4)
    sc.GetObject(1).GetSection(0).SetVisibleToSelf(False)                 # hidden only from its own rays: still occludes the floor
    assert (_ao(orc, sc) == 0.0).all()


def test_section_hidden_from_self_does_not_occlude_itself(orc):
    def covered(sc):
        w = FlatScene(sc)       # keeps the arrays alive
        return w
    sc = scenes.cube_scene(32, inward=True, bpc=4)
    closed = _ao(orc, sc)
    sc2 = scenes.cube_scene(32, inward=True, bpc=4)
    sc2.GetObject(0).GetSection(0).SetVisibleToSelf(False)
    open_ = _ao(orc, sc2)
    sc3 = scenes.cube_scene(32, inward=True, bpc=4)
    sc3.GetObject(0).GetSection(0).SetVisibleToOthers(False)              # no other section: nothing changes
    assert np.array_equal(_ao(orc, sc3), closed)
    assert closed.max() < 0.05 and (open_[closed < 0.05] >= 0.999).mean() > 0.8 and open_.max() == 1.0


def test_material_range_ends_the_rays(orc):
    h = 0.5
    for rng, use, want in ((0.25, 1, 1.0), (0.25, 0, 0.0), (0.0, 1, 0.0)):
        sc = scenes.floor_ceiling_scene(8, height=h, bpc=4)
        sc.GetMaterial(0).SetAOMaxRange(rng)
        assert (_ao(orc, sc, material_ranges=use) == want).all(), (rng, use)
    sc = scenes.floor_ceiling_scene(8, height=h, bpc=4)
    sc.GetMaterial(0).SetAOMaxRange(10.0)                  # only rays within 2.9 degrees of the horizon reach 10 units
    ao = _ao(orc, sc, material_ranges=1)
    assert ao.max() <= 2 / 64 and ao.mean() < 0.01
    # falloff mode: the range replaces the falloff distance, so the closed form of test (3) holds with F = range
    F = 1.0
    sc = scenes.floor_ceiling_scene(8, height=h, bpc=4)
    sc.GetMaterial(0).SetAOMaxRange(F)
    o = opts(orc, sample_count=4096, sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, use_any_hit=0,
             falloff_distance=7.0, normal_fp16=capi.FP16_OFF, material_ranges=1)
    orc.bake(FlatScene(sc).desc, o, threads=4)
    ao = sc.GetMaterial(0).GetAOTexture().as_array()[..., 0]
    c0 = h / F
    assert abs(ao.mean() - (c0 * c0 + (2 * h / F) * (1 - c0))) < 3e-3


# ---- (4) floor next to a long wall ------------------------------------------------------------
def test_floor_wall_form_factor(orc):
    H = 1.0
    sc = scenes.floor_wall_scene(16, wall_height=H, bpc=4)
    o = opts(orc, sample_count=2048, sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, normal_fp16=capi.FP16_OFF)
    orc.bake(FlatScene(sc).desc, o, threads=4)
    ao = sc.GetMaterial(0).GetAOTexture().as_array()[..., 0]
    d = (np.arange(16) + 0.5) / 16 * 2.0
    expect = 0.5 * (1 + d / np.sqrt(d * d + H * H))
    assert np.abs(ao.mean(axis=0) - expect).max() < 4e-3


# ---- quantised accumulation + resolve (ProcessIntersections:87-90, Postprocess:81) ----------
def test_resolve_integer_division(orc):
    res = np.zeros(4, dtype=capi.result_dtype())
    res["total_bits"] = np.int32([100, 0, 51, 7]).view(np.uint32)
    res["weight_bits"] = np.int32([200, 10, 100, 7]).view(np.uint32)
    out = orc.resolve(res, 4, 1, opts(orc, bleed_distance=0), 1, 1)
    # floor(T*255/Wt) written as UNORM8
    assert out.tolist() == [100 * 255 // 200, 0, 51 * 255 // 100, 255]
    out = orc.resolve(res, 4, 1, opts(orc, bleed_distance=0), 4, 4).view(np.float32).reshape(4, 4)
    assert np.allclose(out[:, 0], np.float32([127, 0, 130, 255]) / np.float32(255))
    assert (out[:, 1] == 0).all() and (out[:, 2] == 0).all() and (out[:, 3] == 1).all()


# ---- (8) bleed -------------------------------------------------------------------------------
def test_bleed_single_texel_makes_an_open_disc(orc):
    W = H = 21
    res = np.zeros(W * H, dtype=capi.result_dtype())
    res["total_bits"][10 * W + 10] = 200
    res["weight_bits"][10 * W + 10] = 255
    out = orc.resolve(res, W, H, opts(orc, bleed_distance=5), 1, 1).reshape(H, W)
    yy, xx = np.mgrid[0:H, 0:W]
    d2 = (xx - 10) ** 2 + (yy - 10) ** 2
    assert (out[d2 < 25] == 200).all()          # strict d^2 < B^2: the ring at distance exactly 5 is excluded
    assert (out[d2 >= 25] == 0).all()


def test_bleed_row_aliasing_quirk(orc):
    """PostprocessPixelShader.hlsl:65 reads ResultBuffer[y*W + x] with the RAW x: a covered texel
    at the end of row r bleeds into the START of row r+1 (x = -1 aliases to the previous row)."""
    W, H = 16, 4
    res = np.zeros(W * H, dtype=capi.result_dtype())
    res["total_bits"][1 * W + 15] = 255
    res["weight_bits"][1 * W + 15] = 255
    out = orc.resolve(res, W, H, opts(orc, bleed_distance=5), 1, 1).reshape(H, W)
    assert out[1, 15] == 255 and out[1, 12] == 255
    assert out[2, 0] == 255          # (x=-1, y=2) aliases to index 2*W-1 = (15, 1)
    assert out[0, 0] == 255          # (x=-1, y=2) again: d^2 = 1 + 4
    assert out[3, 8] == 0 and out[1, 9] == 0


def test_bleed_nearest_mode(orc):
    """AOB_BLEED_NEAREST: no aliasing across rows, inclusive radius, the nearest covered texel wins."""
    W, H = 16, 6
    res = np.zeros(W * H, dtype=capi.result_dtype())
    res["total_bits"][1 * W + 15] = 255; res["weight_bits"][1 * W + 15] = 255        # end of row 1
    res["total_bits"][4 * W + 3] = 100; res["weight_bits"][4 * W + 3] = 255
    out = orc.resolve(res, W, H, opts(orc, bleed_distance=3, bleed_mode=capi.BLEED_NEAREST), 1, 1).reshape(H, W)
    assert out[1, 15] == 255 and out[1, 12] == 255 and out[1, 11] == 0               # radius 3 inclusive along the row
    assert out[2, 0] == 0 and out[0, 0] == 0                                         # nothing leaks into the next row's start
    assert out[4, 0] == 100 and out[4, 6] == 100 and out[4, 7] == 0                  # d = 3 reached on both sides
    assert out[2, 5] == 100 and out[2, 4] == 100 and out[1, 5] == 0                  # d^2 = 8 and 5 <= 9; 13 > 9
    # between two sources the nearer one wins; at equal distance the one met first scanning rows top to bottom
    res2 = np.zeros(W * H, dtype=capi.result_dtype())
    for (y, x, v) in ((1, 4, 50), (3, 4, 200), (2, 8, 120)):
        res2["total_bits"][y * W + x] = v; res2["weight_bits"][y * W + x] = 255
    out2 = orc.resolve(res2, W, H, opts(orc, bleed_distance=3, bleed_mode=capi.BLEED_NEAREST), 1, 1).reshape(H, W)
    assert out2[2, 4] == 50                      # tie between (1,4) and (3,4): the upper row is scanned first
    assert out2[2, 7] == 120 and out2[2, 5] == 50 and out2[3, 5] == 200
    # the reference mode on the same cells: exclusive radius and the x-outer scan order
    ref = orc.resolve(res2, W, H, opts(orc, bleed_distance=3), 1, 1).reshape(H, W)
    assert ref[2, 4] == 50 and ref[2, 11] == 0 and out2[2, 11] == 120


def test_two_sections_sum_into_the_same_cells(orc):
    sc = scenes.two_section_scene(32)
    o = opts(orc, sample_count=16)
    flat = FlatScene(sc)
    surfels, sidx = orc.rasterize(flat.desc, 0, o)
    # both sections contribute a surfel to the overlap; section-local indices restart at 0
    n_a = int((sidx[1:] < sidx[:-1]).argmax()) + 1
    assert sidx[0] == 0 and sidx[n_a] == 0 and n_a == 24 * 24
    cells = surfels["texel_y"] * 32 + surfels["texel_x"]
    assert len(set(cells[:n_a].tolist()) & set(cells[n_a:].tolist())) == 16 * 16


# ---- the 4-ary collapse --------------------------------------------------------------------
@pytest.mark.parametrize("make", [lambda: scenes.terrain_scene(grid=48, atlas=64), lambda: scenes.sphere_scene(n=24, atlas=64),
                                  lambda: scenes.cube_scene(64), lambda: scenes.plane_scene(8)])
@pytest.mark.parametrize("pairs", [False, True])
def test_bvh4_is_a_collapse_of_the_bvh2(orc, make, pairs):
    flat = FlatScene(make())
    nodes2, depth = orc.bvh_build(orc.scene_triangles(flat.desc))
    nodes4, tris = orc.bvh4_build(nodes2, pairs=pairs)
    n = (len(nodes2) + 1) // 2
    assert len(tris) == n and len(nodes4) == max(1, n - 1)
    refs = nodes4["lo"][:, :, 3].view(np.uint32)
    # walk from the root: every triangle is reached exactly once, every visited node is non-empty
    seen, stack, visited, n_pairs = [], [0], 0, 0
    while stack:
        i = stack.pop(); visited += 1
        for j in range(4):
            r = int(refs[i, j])
            if r == capi.INVALID_ADDR:
                continue
            lo, hi = nodes4["lo"][i, j, :3], nodes4["hi"][i, j, :3]
            assert (lo <= hi).all()
            if r & 0x80000000:
                p = r & 0x3fffffff
                ps = [p, p + 1] if r & 0x40000000 else [p]          # AOB_PAIR_FLAG: the triangles p and p + 1
                n_pairs += len(ps) - 1
                seen += ps
                v = np.concatenate([np.stack([tris["v0"][q, :3], tris["v1"][q, :3], tris["v2"][q, :3]]) for q in ps])
                assert np.array_equal(lo, v.min(axis=0)) and np.array_equal(hi, v.max(axis=0))
            else:
                stack.append(r)
    assert sorted(seen) == list(range(n))
    assert (n_pairs > 0) == (pairs and n > 2)
    if pairs and n > 2:
        # a pair is a BVH2 node whose children are adjacent leaves, and never a node of its own
        inner = nodes2[:n - 1]
        is_pair = (inner["addr_left"] >= n - 1) & (inner["addr_right"] == inner["addr_left"] + 1)
        is_pair[0] = False
        assert n_pairs == int(is_pair.sum())
        assert (refs[is_pair][:, 0] == 0).all()                          # their own records stay empty
    if n > 1 and not pairs:
        # the boxes are BVH2 boxes: the root's four children are the root's grandchildren
        root2 = nodes2[0]
        kids = []
        for a, lo, hi in ((root2["addr_left"], root2["lmin_or_v0"], root2["lmax_or_v1"]), (root2["addr_right"], root2["rmin_or_v2"], root2["rmax"])):
            if a < n - 1:
                c = nodes2[a]
                kids += [(c["lmin_or_v0"], c["lmax_or_v1"]), (c["rmin_or_v2"], c["rmax"])]
            else:
                kids.append((lo, hi))
        for j, (lo, hi) in enumerate(kids):
            assert np.array_equal(nodes4["lo"][0, j, :3], lo) and np.array_equal(nodes4["hi"][0, j, :3], hi)


@pytest.mark.parametrize("any_hit", [1, 0])
def test_bvh4_and_bvh2_walks_agree(orc, any_hit):
    """The 4-ary walk tests a subset of the BVH2 boxes against the same triangles: same answer."""
    flat = FlatScene(scenes.terrain_scene(grid=48, atlas=64))
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    out = {}
    for width in (2, 4):
        o = opts(orc, sample_count=32, sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, use_any_hit=any_hit,
                 bvh_width=width, seed=9)
        surfels, sidx = orc.rasterize(flat.desc, 0, o)
        out[width], st = orc.trace(nodes, surfels, sidx, o, 64, 64, threads=4)
        out[(width, "visits")] = (st.inner_visits + st.leaf_visits) / st.rays
    assert np.array_equal(out[2]["total_bits"], out[4]["total_bits"])
    assert np.array_equal(out[2]["weight_bits"], out[4]["weight_bits"])
    assert out[(4, "visits")] < 0.7 * out[(2, "visits")]      # and it fetches far fewer nodes
