"""Parity at BASELINE.json's full sizes (the oracle runs on all host cores of the GPU box).

configs[1]: ~70 k-triangle mesh (sphere stand-in for the bunny), 1024^2 atlas, 64 spp — complete bake.
configs[2]: 999 698-triangle terrain, 2048^2 atlas — BVH, surfels, and the trace at a reduced sample
count bit for bit; at the full 256 spp the GPU image is checked through size-independent properties
(determinism, sharded == unsharded) and against the oracle on a texel subsample.
"""
import os


import numpy as np
import pytest

from aobaker_amd import Baker, capi, scenes
from aobaker_amd.scene import FlatScene

pytestmark = pytest.mark.gpu
CORES = os.cpu_count() or 1


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8).tobytes()


def cosine_options(orc, spp, **kw):
    o = orc.default_options()
    o.sample_count, o.sampler, o.accumulate, o.normal_fp16, o.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, 3
    for k, v in kw.items():
        setattr(o, k, v)
    return o


@pytest.fixture(scope="module")
def terrain():
    return scenes.terrain_scene(grid=708, atlas=2048, seed=3, bpc=4)


@pytest.mark.parametrize("mesh", ["stand-in", "torus knot OBJ"])
def test_config2_complete_bake(engine, orc, mesh, tmp_path, monkeypatch):
    """configs[1], every material of it, GPU == oracle.  The mesh is scenes.config2_scene's: the displaced sphere standing in
    for the bunny, or — AOB_CFG2_OBJ=/path/to.obj — the mesh in that file (imported like BakeObject::LoadFromFile).  Second
    parametrisation (round 5): a 70 000-triangle displaced (2, 3) torus knot written as OBJ by scenes.torus_knot_obj and baked THROUGH
    that override — a mesh of the bunny's class that, unlike the sphere, is not star-shaped: concavities, a handle, strands that pass
    over and shadow each other (a quarter of its rays are occluded)."""
    if mesh == "torus knot OBJ" and not os.environ.get("AOB_CFG2_OBJ"):
        monkeypatch.setenv("AOB_CFG2_OBJ", scenes.torus_knot_obj(str(tmp_path / "torus_knot_70k.obj"), 700, 50, seed=2))
    (gpu, name), (cpu, _) = scenes.config2_scene(atlas=1024, seed=2, bpc=4), scenes.config2_scene(atlas=1024, seed=2, bpc=4)
    opt = cosine_options(orc, 64)
    opt.seed = 2
    engine.Options = opt
    engine.Bake(gpu)
    st = orc.bake(FlatScene(cpu).desc, opt, threads=CORES)
    print(f"configs[1] mesh: {name}, {engine.LastStats.triangle_count} triangles, {st.rays} rays")
    assert engine.LastStats.ray_count == st.rays and st.rays > 0
    if "stand-in" in name:
        assert engine.LastStats.triangle_count == 69938
    if "torus_knot_70k" in name:
        assert engine.LastStats.triangle_count == 70000 and 0.15 < st.hits / st.rays < 0.4        # it shadows itself
    for i in range(1, gpu.GetMaterialCount()):
        assert np.array_equal(gpu.GetMaterial(i).GetAOTexture().GetData(), cpu.GetMaterial(i).GetAOTexture().GetData())
    g, c = gpu.GetMaterial(0).GetAOTexture(), cpu.GetMaterial(0).GetAOTexture()
    assert np.abs(g.ao() - c.ao()).max() <= 1e-4
    assert np.array_equal(g.GetData(), c.GetData())
    assert 0.2 < g.ao()[g.ao() > 0].mean() < 1.0


@pytest.mark.parametrize("builder", [1, 0, 2])
def test_config3_bvh_and_surfels_bit_exact(engine, orc, terrain, builder):
    flat = FlatScene(terrain)
    opt = cosine_options(orc, 4, bvh_builder=builder)
    tris = orc.scene_triangles(flat.desc)
    ref_nodes, ref_depth = orc.bvh_build(tris, builder)
    refq, ref_frame = orc.bvh2q_build(ref_nodes)
    ref_surfels, _ = orc.rasterize(flat.desc, 0, opt)
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0)
        st = b.stats()
        assert (st.triangle_count, st.node_count, st.bvh_depth, st.surfel_count) == (999698, 1999395, ref_depth, 2048 * 2048)
        assert bits(b.nodes()) == bits(ref_nodes)          # 1 M-leaf refit: every box, every link
        gq, gf = b.nodes2q()
        assert bits(gf) == bits(ref_frame) and np.array_equal(gq, refq)
        assert bits(b.surfels(0)) == bits(ref_surfels)
        # and the whole trace at 4 spp (16.8 M rays) against the oracle, every cell
        b.trace(0)
        got = b.results(0)
    ref, _ = orc.trace(ref_nodes, ref_surfels, None, opt, 2048, 2048, threads=CORES)
    assert np.array_equal(got["total_bits"], ref["total_bits"]) and np.array_equal(got["weight_bits"], ref["weight_bits"])


@pytest.mark.parametrize("builder", [1, 2])
def test_config3_full_spp_properties(engine, orc, terrain, builder):
    """bench.py's own configuration (bvh_builder 2; 1 is the library's default) at its own 256 spp: determinism, sharded ==
    unsharded, the oracle on a texel subset, and the resolved atlas's CRC-32 — the constant bench.py carries
    (config.image_crc32_expected): an any-hit answer does not depend on the tree, so both builders give the same image."""
    import zlib
    import bench
    flat = FlatScene(terrain)
    opt = cosine_options(orc, 256, bvh_builder=builder)
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0)
        b.trace(0)
        first = b.results(0)
        # determinism: the wave-level work pool and LDS atomics must not leak scheduling into the sums
        b.clear_results(0); b.trace(0)
        assert bits(b.results(0)) == bits(first)
        # sharded (8 tile-row aligned parts, as 8 GPUs would trace them) == unsharded
        bounds, strips, overlapping = b.shard_bounds(0, 8)
        assert not overlapping and bounds[-1] == 2048 * 2048
        b.clear_results(0)
        for r in range(8):
            b.trace(0, bounds[r], bounds[r + 1])
        assert bits(b.results(0)) == bits(first)
        b.resolve(0)
        img = np.zeros(2048 * 2048 * 4, dtype=np.uint8)
        b.download(0, img)
    assert zlib.crc32(img.tobytes()) == bench.CONFIG2_IMAGE_CRC32
    # ... which, since round 6, is the ORACLE's number: tests/golden/make_fullsize_golden.py bakes this atlas completely on the CPU (1.07 G rays) and
    # writes the CRC bench.py reads, the CRC of the cells, and the cells of every 16th texel in x and y — compared here without running the oracle
    import json
    golden = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fullsize_crc.json")))["2"]
    assert golden["image_crc32"] == bench.CONFIG2_IMAGE_CRC32 and golden["surfels"] == 2048 * 2048
    assert zlib.crc32(np.ascontiguousarray(first).tobytes()) == golden["cells_crc32"]
    sub = np.load(os.path.join(os.path.dirname(__file__), "golden", "fullsize_cells_2.npz"))
    assert np.array_equal(first["total_bits"][sub["cell_index"]], sub["total_bits"]) and np.array_equal(first["weight_bits"][sub["cell_index"]], sub["weight_bits"])
    ao = img.view(np.float32).reshape(2048, 2048)
    t = first["total_bits"].view(np.float32).reshape(2048, 2048)
    w = first["weight_bits"].view(np.float32).reshape(2048, 2048)
    assert (w == 256.0).all()                               # unit weights: every texel cast 256 rays
    assert (t == np.round(t)).all() and t.min() >= 0 and t.max() <= 256
    assert np.array_equal(ao, t / w)
    assert 0.5 < ao.mean() < 0.95
    # the oracle on every 32nd texel in x and y (4096 surfels x 256 spp), exact
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc), builder, 1)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    keep = (surfels["texel_x"] % 32 == 0) & (surfels["texel_y"] % 32 == 0)
    sub = np.ascontiguousarray(surfels[keep])
    ref, _ = orc.trace(nodes, sub, None, opt, 2048, 2048, threads=CORES)
    cells = sub["texel_y"].astype(np.int64) * 2048 + sub["texel_x"]
    assert np.array_equal(first["total_bits"][cells], ref["total_bits"][cells])
    assert np.array_equal(first["weight_bits"][cells], ref["weight_bits"][cells])
