"""The oracle's two-level walk (top-level tree over instances + shared bottom-level trees, the structure of
kernels/CL/intersect_bvh2level_skiplinks.cl) finds what the flat walk over the flattened scene finds."""
import numpy as np

from aobaker_amd import capi, scenes
from aobaker_amd.scene import FlatScene


def test_two_level_walk_agrees_with_the_flat_walk(orc):
    sc = scenes.instanced_scene(instances=12, rock_n=20, atlas=128, seed=5)
    flat = FlatScene(sc)
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.seed, opt.normal_fp16 = 16, capi.SAMPLER_COSINE, 5, capi.FP16_OFF
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    rays = orc.generate_rays(surfels[::7], None, opt, 128, t_max=1.25)
    tlas, instances, blas = orc.two_level_scene(sc)
    assert len(instances) == 13 and len(blas) == 2                     # the ground and ONE rock tree shared by 12 instances
    assert len(tlas) == 2 * 13 - 1
    for any_hit in (False, True):
        fh, fst = orc.flat_intersect(nodes, rays, any_hit)
        th, top, bottom = orc.two_level_intersect(tlas, instances, blas, rays, any_hit)
        hit_f, hit_t = fh["shape_id"] != -1, th["shape_id"] != -1
        # object-space arithmetic rounds differently from world-space: a ray grazing an edge may change sides
        assert (hit_f != hit_t).mean() < 2e-3 and hit_f.any()
        both = hit_f & hit_t
        if not any_hit:
            same = both & (fh["shape_id"] == th["shape_id"])
            assert same.sum() > 0.995 * both.sum()
            assert np.abs(fh["uvwt"][same, 3] - th["uvwt"][same, 3]).max() < 1e-4
            assert (fh["prim_id"][same] == th["prim_id"][same]).mean() > 0.995
        assert top.rays == fst.rays == (rays["extra"][:, 1] != 0).sum()
        assert top.leaf_visits > 0 and bottom.inner_visits > 0 and bottom.leaf_visits > 0
