"""Randomised GPU-vs-oracle parity: random scenes (triangle soups with random UVs, bumpy grids, degenerate and duplicate
triangles, several objects / sections / materials of ragged atlas sizes with random transforms, visibility flags and ranges) baked with random
options (every sampler, scoring, accumulation, structure, builder, Morton order, bleed mode, fp16 mode, odd sample counts,
short rays, zero bias, both raster modes, all layers of overlapping charts) must give the oracle's textures byte for byte.  Seeds are fixed: the cases are the same every run."""
import numpy as np
import pytest

from aobaker_amd import capi
from aobaker_amd.scene import BakeObject, FlatScene, MaterialObject, RenderMesh, Scene, Section, make_vertices

pytestmark = pytest.mark.gpu


def random_scene(rng, bpc):
    sc = Scene()
    nmat = int(rng.integers(1, 3))
    mats = [sc.AddMaterial(MaterialObject(f"m{i}", int(rng.integers(1, 100)), int(rng.integers(1, 100)), 1, bpc)) for i in range(nmat)]
    for m in mats:
        m.SetAOMaxRange(float(rng.uniform(0.1, 2.0)))
    for o in range(int(rng.integers(1, 4))):
        obj = sc.AddObject(BakeObject(f"o{o}", position=rng.uniform(-0.3, 0.3, 3), rotation=rng.uniform(-1, 1, 3), scale=rng.uniform(0.5, 1.5, 3)))
        for s in range(int(rng.integers(1, 3))):
            kind = rng.integers(0, 3)
            if kind == 0:      # triangle soup with random uvs
                n = int(rng.integers(1, 400))
                c = rng.uniform(0, 1, (n, 1, 3)); v = (c + rng.normal(scale=rng.uniform(0.02, 0.3), size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
                uvc = rng.uniform(0, 1, (n, 1, 2)); uv = (uvc + rng.normal(scale=0.05, size=(n, 3, 2))).astype(np.float32).reshape(-1, 2)
                nrm = rng.normal(size=(3 * n, 3)).astype(np.float32); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
                idx = np.arange(3 * n, dtype=np.uint32)
            elif kind == 1:    # bumpy grid
                g = int(rng.integers(2, 24))
                x, z = np.meshgrid(np.linspace(0, 1, g + 1), np.linspace(0, 1, g + 1))
                y = 0.1 * np.sin(6 * x + rng.uniform(0, 6)) * np.cos(5 * z)
                v = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
                uv = np.stack([x, z], -1).reshape(-1, 2).astype(np.float32)
                nrm = np.tile(np.float32([0, 1, 0]), (len(v), 1))
                q = np.arange(g * (g + 1)).reshape(g, g + 1)[:, :-1].reshape(-1)
                idx = np.stack([q, q + 1, q + g + 2, q, q + g + 2, q + g + 1], 1).reshape(-1).astype(np.uint32)
            else:              # degenerate stuff: zero-area and duplicate triangles, coincident planes
                v = np.float32([[0, 0, 0], [1, 0, 0], [1, 0, 1], [0, 0, 1], [0.5, 0, 0.5], [0.5, 0, 0.5]])
                uv = np.float32([[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.5], [0.5, 0.5]])
                nrm = np.tile(np.float32([0, 1, 0]), (6, 1))
                idx = np.uint32([0, 1, 2, 0, 2, 3, 0, 1, 2, 4, 5, 4, 0, 0, 0])
            sec = obj.AddSection(Section("s", RenderMesh(make_vertices(v, nrm, uv), idx), mats[int(rng.integers(0, nmat))] if rng.uniform() < 0.9 else None))
            sec.SetVisibleToSelf(bool(rng.uniform() < 0.8)); sec.SetVisibleToOthers(bool(rng.uniform() < 0.8))
    return sc


def random_options(rng):
    o = capi.default_options()
    o.sample_count = int(rng.choice([1, 3, 16, 17, 64]))
    o.sampler = int(rng.integers(0, 2)); o.accumulate = int(rng.integers(0, 2)); o.use_any_hit = int(rng.integers(0, 2))
    o.seed = int(rng.integers(0, 1000)); o.normal_fp16 = int(rng.integers(0, 3)); o.bleed_mode = int(rng.integers(0, 2)); o.bleed_distance = int(rng.integers(0, 7))
    o.falloff_distance = float(rng.uniform(0.1, 2.0)); o.clamp_tmax_to_falloff = int(rng.integers(0, 2)); o.material_ranges = int(rng.integers(0, 2))
    o.ray_tmax = float(rng.choice([1e6, 0.7, 3.0])); o.ray_bias = float(rng.choice([0.001, 0.0, 0.01]))
    o.bvh_width, o.bvh_quantised = [(4, 1), (4, 1), (2, 1), (2, 0)][int(rng.integers(0, 4))]
    o.bvh_builder = int(rng.integers(0, 3)); o.morton_adaptive = int(rng.integers(0, 2))
    o.raster_mode = int(rng.integers(0, 2))       # drawn last: the cases keep the options they had before it existed
    o.raster_layers = int(rng.integers(0, 2))
    o.ray_chunk_size = int(rng.choice([0xFFFFFF, 4096, 640, 64]))      # the compat sampler's chunk-local table index wraps
    if o.raster_layers == capi.LAYERS_ALL:
        o.raster_mode = capi.RASTER_REFERENCE     # the layers are centre-sampled
    # the schedule of the trace kernel (drawn last, like everything added later): it must never show in a texel
    if rng.uniform() < 0.5:
        o.trace_refill_idle = int(rng.choice([0, 1, 7, 24, 42, 64])); o.trace_leaf_wait = int(rng.choice([0, 1, 5, 16, 40, 64]))
        o.trace_parts = int(rng.choice([0, 1, 3, 8])); o.trace_part_min = int(rng.choice([0, 1, 4, 16])); o.trace_fine_div = int(rng.choice([0, 1, 3, 8]))
        o.trace_bands = int(rng.integers(0, 3))
        o.trace_scalar_min = int(rng.choice([0, 1, 8, 24, 64]))
        o.trace_cut = int(rng.choice([0, 0, 1, 4, 7, 16, 32]))
    # horizon tables (round 6; drawn after everything else): forced on — they apply to the COSINE sampler's full grids (16, 64 samples here) on the quantised 4-ary nodes
    o.trace_horizon = int(rng.choice([0, 1, 2, 2]))
    return o


def flattened_scene(rng, bpc, axes):
    """random_scene with every object squashed onto the origin along `axes` (scale 0, no rotation, no offset there): a scene
    box without extent on those axes — a quantisation frame whose scale is 0 there, and with all three a scene that is one
    point (every triangle degenerate, every empty child slot indistinguishable from a box: bvh_build.hip k_emit_nodes4q)."""
    sc = random_scene(rng, bpc)
    for obj in sc.BakeObjects:
        obj.Rotation = np.zeros(3, dtype=np.float32)
        for a in axes:
            obj.Scale[a] = 0.0
            obj.Position[a] = 0.0
    return sc


def bake_and_compare(engine, orc, seeds, make_scene=random_scene):
    for seed in seeds:
        rng = np.random.default_rng(seed)
        o = random_options(rng)
        bpc = 1 if o.accumulate == capi.ACCUMULATE_QUANTISED else 4
        gpu_scene, cpu_scene = make_scene(np.random.default_rng(seed), bpc), make_scene(np.random.default_rng(seed), bpc)
        engine.Options = o
        engine.Bake(gpu_scene)
        orc.bake(FlatScene(cpu_scene).desc, o, threads=8)
        for i in range(gpu_scene.GetMaterialCount()):
            g, c = gpu_scene.GetMaterial(i).GetAOTexture().GetData(), cpu_scene.GetMaterial(i).GetAOTexture().GetData()
            assert np.array_equal(g, c), (seed, i, {n: getattr(o, n) for n, _ in capi.AobOptions._fields_ if n != "reserved"})


@pytest.mark.parametrize("block", range(8))
def test_random_scenes_and_options(engine, orc, block):
    bake_and_compare(engine, orc, range(1000 + 50 * block, 1000 + 50 * (block + 1)))


@pytest.mark.parametrize("axes", [(1,), (0, 2), (0, 1, 2)])
def test_random_scenes_without_extent(engine, orc, axes):
    """Flat, line-like and point-like scenes (the scene box has no extent on one, two or all three axes), random options."""
    bake_and_compare(engine, orc, range(3000, 3040), lambda rng, bpc: flattened_scene(rng, bpc, axes))


if __name__ == "__main__":
    # soak: PYTHONPATH=. python tests/test_gpu_fuzz.py FIRST COUNT   (seeds beyond the 400 the suite runs; same generator)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from aobaker_amd import BakeEngine
    from oracle import oracle
    first, count = int(sys.argv[1]), int(sys.argv[2])
    BakeEngine.Init(0)
    for lo in range(first, first + count, 250):
        bake_and_compare(BakeEngine, oracle, range(lo, min(lo + 250, first + count)))
        print(f"seeds {lo}..{min(lo + 250, first + count) - 1}: identical", flush=True)
    BakeEngine.Options = None
    BakeEngine.Dispose()
