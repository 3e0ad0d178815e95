"""Tile cuts (trace.hip CUT mode; oracle orc_tile_cut / orc_cut_walk): a ray that starts below ANY cut of the quantised 4-ary
tree reaches exactly the triangles the walk from the root reaches — child boxes are nested on the quantisation grid and the
slab test is monotone in the planes — so its any-hit answer is the root walk's, ray for ray.  CPU only."""
import numpy as np
import pytest

from aobaker_amd import capi, scenes
from aobaker_amd.scene import FlatScene


def _structures(orc, flat, builder):
    tris = orc.scene_triangles(flat.desc)
    nodes2 = orc.bvh_build(tris, builder, 1)[0]
    n4, lt = orc.bvh4_build(nodes2, pairs=True)
    _, frame = orc.bvh2q_build(nodes2)
    return nodes2, orc.bvh4q_build(n4, frame), frame, lt


def _options(orc, spp, seed):
    o = orc.default_options()
    o.sample_count, o.sampler, o.accumulate, o.normal_fp16, o.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, seed
    return o


SCENES = {
    "terrain": lambda: scenes.terrain_scene(grid=48, atlas=64, seed=3),
    "sphere": lambda: scenes.sphere_scene(n=24, atlas=64, seed=2),
    "soup": lambda: scenes.soup_scene(seed=11, n=300, atlas=64),
    "cube_inward": lambda: scenes.cube_scene(64, inward=True),
}


@pytest.mark.parametrize("name", sorted(SCENES))
@pytest.mark.parametrize("builder", [0, 1, 2])
def test_walk_below_a_cut_equals_walk_from_the_root(orc, name, builder):
    flat = FlatScene(SCENES[name]())
    opt = _options(orc, 16, 5)
    nodes2, n4q, frame, lt = _structures(orc, flat, builder)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    assert len(surfels) > 64
    rng = np.random.default_rng(3)
    for t in rng.choice(len(surfels) // 64, size=min(12, len(surfels) // 64), replace=False):
        sub = np.ascontiguousarray(surfels[64 * t:64 * t + 64])
        rays = np.ascontiguousarray(orc.generate_rays(sub, None, opt, 64))
        plain, _ = orc.walk4q(n4q, frame, lt, rays, True)
        pos = sub["position"].astype(np.float32)
        boxes = [(pos.min(axis=0) - np.float32(1e-3), pos.max(axis=0) + np.float32(1e-3)),             # the tile's own box
                 (np.float32([-1e9] * 3), np.float32([1e9] * 3)),                                      # everything meets it: breadth first
                 (np.float32([5e8] * 3), np.float32([1e9] * 3))]                                       # nothing does: the root's slots
        for lo, hi in boxes:
            for rule in (0, 1):
                for E in (4, 7, 16, 32):
                    cut = orc.tile_cut(n4q, frame, lo, hi, E, rule)
                    assert 1 <= len(cut) <= max(E, 4)
                    st, hits = orc.cut_walk(n4q, frame, lt, rays, cut)
                    assert st.rays == plain.rays and st.hits == plain.hits, (name, builder, rule, E)
                    # ray for ray: the root walk's answer
                    for i in range(0, len(rays), 257):                      # a sample of single rays through the same counter
                        a, _ = orc.walk4q(n4q, frame, lt, rays[i:i + 1], True)
                        assert int(a.hits) == int(hits[i]), (name, builder, rule, E, i)


def test_a_cut_is_a_cut(orc):
    """Every root-to-leaf path of the tree crosses exactly one entry: the triangles below the entries partition the tree's."""
    flat = FlatScene(scenes.terrain_scene(grid=40, atlas=32, seed=3))
    nodes2, n4q, frame, lt = _structures(orc, flat, 2)
    n4q = np.ascontiguousarray(n4q, dtype=np.uint32).reshape(-1, 16)

    def leaves(ref, out):
        if ref & 0x80000000:
            p = int(ref & 0x3fffffff)
            out.append(p)
            if ref & 0x40000000:
                out.append(p + 1)
            return
        for j in range(4):
            w = n4q[ref, 4 * j:4 * j + 4]
            if (w[:3] == 0x8000ffff).all():
                continue
            leaves(int(w[3]), out)

    everything = []
    leaves(0, everything)
    assert sorted(everything) == list(range(len(lt)))
    for lo, hi in ((np.float32([0.4, -1, 0.4]), np.float32([0.45, 1, 0.45])), (np.float32([-1e9] * 3), np.float32([1e9] * 3))):
        for E in (4, 9, 32):
            cut = orc.tile_cut(n4q, frame, lo, hi, E, 1)
            below = []
            for ref in cut[:, 0]:
                leaves(int(ref), below)
            assert sorted(below) == list(range(len(lt))), E
