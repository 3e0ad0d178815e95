"""The oracle's restatement of the reference's own records — 48-byte rays in, 32-byte Intersections out — agrees with
its fused trace (CPU only; tests/test_gpu_reference_kernels.py feeds the same records to the reference's kernels)."""
import numpy as np
import pytest

from aobaker_amd import capi, scenes
from aobaker_amd.scene import FlatScene


def make_options(orc, **kw):
    o = orc.default_options()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


@pytest.mark.parametrize("mode", ["compat_quantised_falloff", "cosine_float_anyhit", "cosine_float_falloff"])
@pytest.mark.parametrize("name", ["cube_inward", "two_section", "sphere"])
def test_record_path_equals_fused_trace(orc, name, mode):
    sc = {"cube_inward": lambda: scenes.cube_scene(24, inward=True), "two_section": lambda: scenes.two_section_scene(24),
          "sphere": lambda: scenes.sphere_scene(n=16, atlas=32)}[name]()
    flat = FlatScene(sc)
    kw = dict(sample_count=9, seed=7, bvh_width=2, bvh_quantised=0)
    if mode == "compat_quantised_falloff":
        kw.update(sampler=capi.SAMPLER_REFERENCE_COMPAT, use_any_hit=0, accumulate=capi.ACCUMULATE_QUANTISED)
    elif mode == "cosine_float_anyhit":
        kw.update(sampler=capi.SAMPLER_COSINE, use_any_hit=1, accumulate=capi.ACCUMULATE_FLOAT)
    else:
        kw.update(sampler=capi.SAMPLER_COSINE, use_any_hit=0, accumulate=capi.ACCUMULATE_FLOAT, falloff_distance=0.6)
    opts = make_options(orc, **kw)
    m = flat.material_array[0]
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    surfels, sidx = orc.rasterize(flat.desc, 0, opts)
    table = orc.compat_table(opts) if opts.sampler == capi.SAMPLER_REFERENCE_COMPAT else None
    want, st = orc.trace(nodes, surfels, sidx, opts, m.width, m.height, hemisphere=table)
    starts = [i for i in range(len(sidx)) if sidx[i] == 0] + [len(sidx)]
    got = np.zeros(m.width * m.height, dtype=capi.result_dtype())
    hits_total = 0
    for sb, se in zip(starts[:-1], starts[1:]):
        rays = orc.generate_rays(surfels, sidx, opts, m.width, hemisphere=table, begin=sb, end=se)
        assert rays.dtype.itemsize == 48 and len(rays) == (se - sb) * 9
        assert (rays["o"][:, 3] == np.float32(opts.ray_tmax)).all() and (rays["extra"][:, 1] == np.int32(-842150451)).all()
        hits = orc.intersect_rays(nodes, rays)
        occ = orc.occluded_rays(nodes, rays)
        assert np.array_equal(occ == 1, hits["shape_id"] != -1)
        hits_total += int((occ == 1).sum())
        orc.process_intersections(surfels[sb:se], rays, hits, opts, m.width, m.height, results=got)
    assert hits_total == st.hits
    assert got.tobytes() == want.tobytes()


def test_barycentrics_locate_the_hit_point(orc):
    """uvwt of a hit (intersect_bvh2_lds.cl:207-217): o + t d == (1-u-v) v1 + u v2 + v v3."""
    flat = FlatScene(scenes.cube_scene(16, inward=True))
    opts = make_options(orc, sample_count=8, sampler=capi.SAMPLER_COSINE, seed=1)
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc), 0, 0)
    surfels, sidx = orc.rasterize(flat.desc, 0, opts)
    rays = orc.generate_rays(surfels, sidx, opts, 16)
    hits = orc.intersect_rays(nodes, rays)
    tris = {(int(t["shape_id"]), int(t["prim_id"])): t["v"] for t in orc.scene_triangles(flat.desc)}
    assert (hits["shape_id"] != -1).all()
    for r, h in list(zip(rays, hits))[::37]:
        v = tris[(int(h["shape_id"]), int(h["prim_id"]))].astype(np.float64)
        u, w, t = float(h["uvwt"][0]), float(h["uvwt"][1]), float(h["uvwt"][3])
        p = r["o"][:3].astype(np.float64) + t * r["d"][:3].astype(np.float64)
        q = (1 - u - w) * v[0] + u * v[1] + w * v[2]
        assert np.abs(p - q).max() < 1e-5


def test_morton_codes_top_bits_are_the_reference_code(orc):
    """build_hlbvh.cl:69-89 in numpy on the oracle's boxes == the top 30 bits of the oracle's 63-bit code."""
    flat = FlatScene(scenes.terrain_scene(grid=24, atlas=16))
    tris = orc.scene_triangles(flat.desc)
    codes, boxes, scene = orc.morton_codes(tris, adaptive=0)
    c = ((boxes[:, 4:7] + boxes[:, 0:3]) * np.float32(0.5) - scene[0:3]) / (scene[4:7] - scene[0:3])
    q = np.minimum(np.maximum(c * np.float32(1024.0), np.float32(0)), np.float32(1023)).astype(np.uint32)

    def expand(v):
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v
    ref = expand(q[:, 0]) * np.uint32(4) + expand(q[:, 1]) * np.uint32(2) + expand(q[:, 2])
    assert np.array_equal(ref.astype(np.uint64), codes >> np.uint64(33))
