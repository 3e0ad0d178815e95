"""GPU: the horizon tables (aobaker_amd/csrc/horizon.hip, AobOptions::trace_horizon) against the oracle, through the C ABI.

The tables are an accelerator: a bake with them must write the cells a bake without them writes — bit for bit, the oracle's cells — and every cell
of the sampler's grid they skip must hold only rays the oracle finds unoccluded.  (The kernel's logic on the CPU: tests/test_horizon_emu.py.)"""
import numpy as np
import pytest

from aobaker_amd import Baker, capi, scenes
from aobaker_amd.scene import FlatScene

pytestmark = pytest.mark.gpu

SCENES = {
    "cube": lambda: scenes.cube_scene(64),
    "cube_inward": lambda: scenes.cube_scene(64, inward=True),
    "two_section": lambda: scenes.two_section_scene(32),
    "floor_wall": lambda: scenes.floor_wall_scene(32),
    "sphere": lambda: scenes.sphere_scene(n=48, atlas=128),
    "terrain": lambda: scenes.terrain_scene(grid=96, atlas=128),
    "terrain_fine": lambda: scenes.terrain_scene(grid=200, atlas=256, seed=7),
    "soup": lambda: scenes.soup_scene(atlas=96),
    "rocks": lambda: scenes.instanced_scene(instances=9, rock_n=24, atlas=128),
}

MODES = [
    dict(use_any_hit=1, accumulate=capi.ACCUMULATE_FLOAT),
    dict(use_any_hit=1, accumulate=capi.ACCUMULATE_QUANTISED),
    dict(use_any_hit=0, accumulate=capi.ACCUMULATE_FLOAT, falloff_distance=0.4, clamp_tmax_to_falloff=1),
    dict(use_any_hit=0, accumulate=capi.ACCUMULATE_QUANTISED, falloff_distance=1.25, clamp_tmax_to_falloff=0),
    dict(use_any_hit=1, accumulate=capi.ACCUMULATE_FLOAT, ray_tmax=0.3),
]


def options(orc, spp, **kw):
    o = orc.default_options()
    o.sample_count, o.sampler, o.seed = spp, capi.SAMPLER_COSINE, 5
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def bake_cells(flat, opt, horizon):
    opt.trace_horizon = horizon
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0)
        cells, st = b.results(0).copy(), b.stats()
        tables = b.horizon(0) if horizon != 1 else None
        surfels = b.surfels(0)
    return cells, st, tables, surfels


@pytest.mark.parametrize("mode", range(len(MODES)))
@pytest.mark.parametrize("name,spp", [("cube", 16), ("cube", 256), ("cube_inward", 64), ("two_section", 64), ("floor_wall", 256), ("sphere", 64), ("terrain", 64),
                                      ("terrain_fine", 256), ("soup", 64), ("rocks", 64), ("terrain", 1024)])
def test_cells_with_tables_are_the_cells_without_and_the_oracles(engine, orc, name, spp, mode):
    flat = FlatScene(SCENES[name]())
    opt = options(orc, spp, **MODES[mode])
    with_t, st, tables, surfels = bake_cells(flat, opt, 2)
    without, st0, _, _ = bake_cells(flat, opt, 1)
    assert st.trace_horizon == 1 and st0.trace_horizon == 0
    assert st0.rays_cast == st0.ray_count and st.rays_cast <= st.ray_count
    assert np.array_equal(with_t["total_bits"], without["total_bits"]) and np.array_equal(with_t["weight_bits"], without["weight_bits"])
    m = flat.material_array[0]
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    osurf, sidx = orc.rasterize(flat.desc, 0, opt)
    ref, _ = orc.trace(nodes, osurf, sidx, opt, m.width, m.height, None, threads=8)
    assert np.array_equal(with_t["total_bits"], ref["total_bits"]) and np.array_equal(with_t["weight_bits"], ref["weight_bits"])
    # the tables themselves: the samples they skip are the ones the kernel did not cast ...
    sx = int(np.floor(np.sqrt(spp))); sy = spp // sx
    assert tables.shape[0] == len(surfels)
    free = np.minimum(tables[:, :sy].astype(np.int64), sx)
    assert int(free.sum()) == st.ray_count - st.rays_cast
    # ... and every one of them is unoccluded for the oracle (rays of unbounded reach: a superset of any bounded mode's hits)
    if opt.use_any_hit and opt.ray_tmax > 1e5:
        rays = orc.generate_rays(surfels, None, opt, m.width)
        occ = (orc.occluded_rays(nodes, rays).reshape(len(surfels), spp) == 1).reshape(len(surfels), sy, sx)
        skipped = np.arange(sx)[None, None, :] < free[:, :, None]
        assert int((skipped & occ).sum()) == 0


def test_open_sky_is_skipped_and_a_closed_room_is_not(engine, orc):
    opt = options(orc, 256, use_any_hit=1, accumulate=capi.ACCUMULATE_FLOAT)
    _, st, _, _ = bake_cells(FlatScene(scenes.cube_scene(64)), opt, 2)
    assert st.rays_cast < 0.4 * st.ray_count            # a convex body: nothing but the most grazing cells is left to cast
    _, st, _, _ = bake_cells(FlatScene(scenes.cube_scene(64, inward=True)), opt, 2)
    assert st.rays_cast == st.ray_count


def test_the_default_uses_tables_from_128_samples_on(engine, orc):
    flat = FlatScene(scenes.terrain_scene(grid=96, atlas=128))
    for spp, want in ((64, 0), (100, 0), (144, 0), (256, 1), (257, 0), (1024, 1)):      # (144 = 12 x 12: three rows a part do not fit a table word)
        _, st, _, _ = bake_cells(flat, options(orc, spp, use_any_hit=1, accumulate=capi.ACCUMULATE_FLOAT), 0)
        assert st.trace_horizon == want, (spp, st.trace_horizon)
    o = options(orc, 256, use_any_hit=1)
    o.sampler = capi.SAMPLER_REFERENCE_COMPAT
    _, st, _, _ = bake_cells(flat, o, 2)
    assert st.trace_horizon == 0                         # the table sampler's directions know no grid
    for w, q in ((2, 1), (2, 0)):
        _, st, _, _ = bake_cells(flat, options(orc, 256, use_any_hit=1, bvh_width=w, bvh_quantised=q), 2)
        assert st.trace_horizon == 0


@pytest.mark.parametrize("parts", [2, 3])
def test_row_windows_and_ranges_with_tables_stitch_to_the_full_bake(engine, orc, parts):
    flat = FlatScene(scenes.terrain_scene(grid=96, atlas=128))
    opt = options(orc, 256, use_any_hit=1, accumulate=capi.ACCUMULATE_FLOAT)
    full, _, _, _ = bake_cells(flat, opt, 2)
    opt.trace_horizon = 2
    H = flat.material_array[0].height
    acc = np.zeros_like(full)
    for i in range(parts):
        r0, r1 = H * i // parts // 8 * 8, (H * (i + 1) // parts // 8 * 8 if i + 1 < parts else H)
        with Baker(flat, opt) as b:
            b.build_bvh(); b.rasterize(0, rows=(r0, r1)); b.trace(0)
            part = b.results(0)
        acc["total_bits"] |= part["total_bits"]; acc["weight_bits"] |= part["weight_bits"]
    assert np.array_equal(acc["total_bits"], full["total_bits"]) and np.array_equal(acc["weight_bits"], full["weight_bits"])
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0)
        n = b.surfel_count(0)
        for i in range(parts):
            b.trace(0, n * i // parts // 64 * 64, n * (i + 1) // parts // 64 * 64 if i + 1 < parts else n)
        ranged = b.results(0)
    assert np.array_equal(ranged["total_bits"], full["total_bits"]) and np.array_equal(ranged["weight_bits"], full["weight_bits"])


def test_sections_with_visibility_flags_and_a_material_range(engine, orc):
    from tests.test_gpu_parity import _flagged_scene
    for mode in (0, 2):
        for mat_range in (0, 1):
            opt = options(orc, 256, material_ranges=mat_range, **MODES[mode])
            gpu_scene, cpu_scene = _flagged_scene(4), _flagged_scene(4)
            opt.trace_horizon = 2
            engine.Options = opt
            engine.Bake(gpu_scene)
            orc.bake(FlatScene(cpu_scene).desc, opt, threads=8)
            for i in range(gpu_scene.GetMaterialCount()):
                assert np.array_equal(gpu_scene.GetMaterial(i).GetAOTexture().GetData(), cpu_scene.GetMaterial(i).GetAOTexture().GetData())
    engine.Options = None


@pytest.mark.parametrize("parts,want", [(1, 1), (2, 1), (4, 1), (8, 1), (16, 1)])
def test_units_of_one_to_eight_rows(engine, orc, parts, want):
    """256 spp = 16 rows: two parts are units of eight rows (two words of a surfel's table), sixteen parts of one row; one part would be sixteen rows: the schedule makes it four parts (schedule_parts)"""
    flat = FlatScene(scenes.terrain_scene(grid=96, atlas=128))
    opt = options(orc, 256, use_any_hit=1, accumulate=capi.ACCUMULATE_FLOAT, trace_parts=parts, trace_part_min=1)
    with_t, st, _, _ = bake_cells(flat, opt, 2)
    without, _, _, _ = bake_cells(flat, opt, 1)
    assert st.trace_horizon == want
    assert np.array_equal(with_t["total_bits"], without["total_bits"]) and np.array_equal(with_t["weight_bits"], without["weight_bits"])
    if want:
        assert st.rays_cast < st.ray_count
