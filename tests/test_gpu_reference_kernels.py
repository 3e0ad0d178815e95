"""The pin: the REFERENCE's own OpenCL kernels, compiled for gfx950 from the sources under /root/reference
(oracle/ref_kernels.mk) and run on the GPU through the HIP module API (tests/refkernels.py), against

  * the oracle's restatement of them (oracle/aob_oracle.c: traverse / intersect_triangle / intersect_bbox /
    the LBVH build) — bit for bit with the exact-division build, within BASELINE.json's 1e-4 with the default
    (native_recip) build;
  * the library's own kernels: the node array `aob_bake_build_bvh` makes IS the reference's `bvh_node[]`
    (intersect_bvh2_lds.cl:40-52), so the reference's traversal kernel walks the library's tree directly, and the
    result cells made from ITS Intersection records are compared with the cells `k_trace` writes.

What is pinned this way: SURVEY.md §8 rows a2 (build_hlbvh.cl:101-314 — Morton codes, hierarchy, refit) and a6
(intersect_bvh2_lds.cl:65-363 with common.cl:177-246); through the scoring of the reference's records, a5 + a7 as the
oracle states them.  Not pinned here (no runnable reference exists for them): the D3D11 rasteriser and the HLSL resolve.
"""
import ctypes as C

import numpy as np
import pytest

from aobaker_amd import Baker, capi, scenes
from aobaker_amd.scene import FlatScene

import refkernels as rk

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def make_options(orc, **kw):
    o = orc.default_options()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


SCENES = {
    "cube": lambda: scenes.cube_scene(48),
    "cube_inward": lambda: scenes.cube_scene(32, inward=True),
    "sphere": lambda: scenes.sphere_scene(n=48, atlas=96),
    "terrain": lambda: scenes.terrain_scene(grid=96, atlas=96),
    "floor_wall": lambda: scenes.floor_wall_scene(32),
    "two_section": lambda: scenes.two_section_scene(32),
}


def library_tree_and_surfels(flat, opts):
    with Baker(flat, opts) as b:
        b.build_bvh()
        b.rasterize(0)
        return b.nodes(), b.surfels(0)


def filled(dtype, n, byte=0x5A):
    """Records pre-filled with a pattern: a kernel that skips a ray must leave its record alone."""
    return np.frombuffer(bytes([byte]) * (dtype.itemsize * max(1, n)), dtype=dtype).copy()


@pytest.mark.parametrize("builder", [0, 1, 2])
@pytest.mark.parametrize("name", ["cube", "cube_inward", "sphere", "terrain", "floor_wall"])
def test_reference_kernels_intersect_main_matches_oracle(engine, orc, name, builder):
    """intersect_main (exact-division build) on the library's node array — the reference's LBVH (0), PLOC (1) and PLOC
    under the binned-SAH top (2, what bench.py bakes with): shape id, primitive id, barycentrics and t of every ray equal
    the oracle's traverse() — bit for bit."""
    flat = FlatScene(SCENES[name]())
    opts = make_options(orc, sample_count=16, sampler=capi.SAMPLER_COSINE, seed=5, bvh_builder=builder, bvh_width=2, bvh_quantised=0)
    nodes, surfels = library_tree_and_surfels(flat, opts)
    m = flat.material_array[0]
    rays = orc.generate_rays(surfels, None, opts, m.width)
    assert len(rays) == len(surfels) * 16 and (rays["extra"][:, 1] != 0).any()
    fill = filled(orc.INTERSECTION, len(rays))
    with rk.Module("intersect_bvh2_lds_exact.co") as mod:
        got = rk.intersect(mod, nodes, rays, fill)
    want = orc.intersect_rays(nodes, rays, fill=fill)
    hit = want["shape_id"] != -1
    assert hit.any() or name == "cube"          # the outward cube sees nothing; everything else has occluders
    assert np.array_equal(got["shape_id"], want["shape_id"])
    assert np.array_equal(got["prim_id"], want["prim_id"])
    # uvwt is written on a hit only (intersect_bvh2_lds.cl:215-217); a miss leaves the fill pattern
    assert bits(got["uvwt"]).tobytes() == bits(want["uvwt"]).tobytes()


@pytest.mark.parametrize("name", ["cube_inward", "sphere", "terrain", "floor_wall"])
def test_reference_kernels_occluded_main_matches_oracle_and_closest_hit(engine, orc, name):
    """occluded_main == the oracle's any-hit walk == (intersect_main found something): the early-out changes no answer."""
    flat = FlatScene(SCENES[name]())
    opts = make_options(orc, sample_count=12, sampler=capi.SAMPLER_COSINE, seed=9, bvh_width=2, bvh_quantised=0)
    nodes, surfels = library_tree_and_surfels(flat, opts)
    rays = orc.generate_rays(surfels, None, opts, flat.material_array[0].width)
    with rk.Module("intersect_bvh2_lds_exact.co") as mod:
        occ = rk.occluded(mod, nodes, rays, filled(np.dtype(np.int32), len(rays)))
        closest = rk.intersect(mod, nodes, rays, filled(orc.INTERSECTION, len(rays)))
    want = orc.occluded_rays(nodes, rays, fill=filled(np.dtype(np.int32), len(rays)))
    assert np.array_equal(occ, want)
    active = rays["extra"][:, 1] != 0
    assert np.array_equal(occ[active] == 1, closest["shape_id"][active] != -1)


@pytest.mark.parametrize("mode", ["anyhit_float", "falloff_float", "anyhit_quantised", "compat_quantised"])
@pytest.mark.parametrize("name", ["cube_inward", "sphere", "terrain", "two_section"])
def test_reference_kernels_cells_match_the_library(engine, orc, name, mode):
    """The whole trace stage against the reference's device code: rays as GenerateRaysComputeShader.hlsl writes them →
    the reference's intersect_main on the library's tree → scored as ProcessIntersectionsComputeShader.hlsl scores them;
    the cells equal what k_trace accumulates — for the reference's node layout and for the default quantised 4-ary one."""
    flat = FlatScene(SCENES[name]())
    kw = dict(sample_count=16, seed=3)
    if mode == "anyhit_float":
        kw.update(sampler=capi.SAMPLER_COSINE, use_any_hit=1, accumulate=capi.ACCUMULATE_FLOAT)
    elif mode == "falloff_float":
        kw.update(sampler=capi.SAMPLER_COSINE, use_any_hit=0, accumulate=capi.ACCUMULATE_FLOAT, falloff_distance=0.4)
    elif mode == "anyhit_quantised":
        kw.update(sampler=capi.SAMPLER_COSINE, use_any_hit=1, accumulate=capi.ACCUMULATE_QUANTISED)
    else:
        kw.update(sampler=capi.SAMPLER_REFERENCE_COMPAT, use_any_hit=0, accumulate=capi.ACCUMULATE_QUANTISED)
    m = flat.material_array[0]
    cells = {}
    for label, width, quant in (("bvh2_fp32", 2, 0), ("bvh4_quantised", 4, 1)):
        opts = make_options(orc, bvh_width=width, bvh_quantised=quant, **kw)
        with Baker(flat, opts) as b:
            b.build_bvh()
            b.rasterize(0)
            b.trace(0)
            cells[label] = b.results(0)
            nodes, surfels = b.nodes(), b.surfels(0)
    opts = make_options(orc, bvh_width=2, bvh_quantised=0, **kw)
    _, sidx = orc.rasterize(flat.desc, 0, opts)
    table = orc.compat_table(opts) if opts.sampler == capi.SAMPLER_REFERENCE_COMPAT else None
    # one section at a time, as the reference dispatches them (BakeEngine.cpp:509-675)
    starts = [i for i in range(len(sidx)) if sidx[i] == 0] + [len(sidx)]
    want = np.zeros(m.width * m.height, dtype=capi.result_dtype())
    with rk.Module("intersect_bvh2_lds_exact.co") as mod:
        for sb, se in zip(starts[:-1], starts[1:]):
            rays = orc.generate_rays(surfels, sidx, opts, m.width, hemisphere=table, begin=sb, end=se)
            hits = rk.intersect(mod, nodes, rays, filled(orc.INTERSECTION, len(rays)))
            orc.process_intersections(surfels[sb:se], rays, hits, opts, m.width, m.height, results=want)
    assert bits(cells["bvh2_fp32"]).tobytes() == bits(want).tobytes()
    assert bits(cells["bvh4_quantised"]).tobytes() == bits(want).tobytes()


def _large_terrain():
    return scenes.terrain_scene(grid=230, atlas=128, seed=4)          # 2 * 229^2 = 104 882 triangles


@pytest.mark.parametrize("builder", [0, 2])
def test_reference_kernels_on_a_100k_triangle_tree(engine, orc, builder):
    """The same two checks where the tree is no toy: 104 882 triangles (the build's kernels run hundreds of workgroups,
    builder 2's top is cut by the multi-workgroup and the one-workgroup kernels), 4 rays per texel of a 128^2 atlas.
    intersect_main on the library's nodes == the oracle's walk of the oracle's own tree, record for record; and the cells
    scored from the reference kernel's records == the cells k_trace accumulates on the quantised 4-ary form of that tree."""
    flat = FlatScene(_large_terrain())
    m = flat.material_array[0]
    kw = dict(sample_count=4, sampler=capi.SAMPLER_COSINE, seed=8, bvh_builder=builder, use_any_hit=0, falloff_distance=0.3,
              accumulate=capi.ACCUMULATE_FLOAT)
    opts2 = make_options(orc, bvh_width=2, bvh_quantised=0, **kw)
    nodes, surfels = library_tree_and_surfels(flat, opts2)
    tris = orc.scene_triangles(flat.desc)
    assert len(tris) >= 100_000
    ref_nodes, _ = orc.bvh_build(tris, builder, 1)
    assert bits(nodes).tobytes() == bits(ref_nodes).tobytes()
    rays = orc.generate_rays(surfels, None, opts2, m.width)
    fill = filled(orc.INTERSECTION, len(rays))
    with rk.Module("intersect_bvh2_lds_exact.co") as mod:
        got = rk.intersect(mod, nodes, rays, fill)
    want = orc.intersect_rays(ref_nodes, rays, fill=fill)
    assert (want["shape_id"] != -1).mean() > 0.05
    assert np.array_equal(got["shape_id"], want["shape_id"]) and np.array_equal(got["prim_id"], want["prim_id"])
    assert bits(got["uvwt"]).tobytes() == bits(want["uvwt"]).tobytes()
    cells = orc.process_intersections(surfels, rays, got, opts2, m.width, m.height)
    with Baker(flat, make_options(orc, bvh_width=4, bvh_quantised=1, **kw)) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0)
        lib_cells = b.results(0)
    assert bits(lib_cells).tobytes() == bits(cells).tobytes()


@pytest.mark.parametrize("builder", [0, 1, 2])
def test_reference_kernels_lds_stack_spills_to_its_global_stack(engine, orc, builder):
    """intersect_main keeps 15 deferred nodes per work item in LDS and moves them to its global stack when a 16th comes
    (intersect_bvh2_lds.cl:144-153), bringing them back when the LDS part runs empty (:186-196).  The hairball — 6 000
    long thin triangles through one point — makes rays defer 16 and more nodes on every tree the library builds (the
    oracle counts the high-water mark of each ray), and stays below the 45 the reference's 32-word stack can hold: the
    records still equal the oracle's, and occluded_main's early-out (:228-363, the same spill code) changes no answer."""
    from test_gpu_parity import _hairball_scene
    flat = FlatScene(_hairball_scene(4, n=6000))
    opts = make_options(orc, sample_count=16, sampler=capi.SAMPLER_COSINE, seed=5, bvh_builder=builder, bvh_width=2, bvh_quantised=0)
    nodes, surfels = library_tree_and_surfels(flat, opts)
    rays = orc.generate_rays(surfels, None, opts, flat.material_array[0].width)
    depth = orc.reference_stack_depths(nodes, rays)
    assert (depth >= 16).sum() >= 32 and depth.max() <= 45, (int((depth >= 16).sum()), int(depth.max()))
    fill = filled(orc.INTERSECTION, len(rays))
    with rk.Module("intersect_bvh2_lds_exact.co") as mod:
        got = rk.intersect(mod, nodes, rays, fill)
        occ = rk.occluded(mod, nodes, rays, filled(np.dtype(np.int32), len(rays)))
    want = orc.intersect_rays(nodes, rays, fill=fill)
    deep = depth >= 16
    assert (want["shape_id"][deep] != -1).any()
    assert np.array_equal(got["shape_id"], want["shape_id"]) and np.array_equal(got["prim_id"], want["prim_id"])
    assert bits(got["uvwt"]).tobytes() == bits(want["uvwt"]).tobytes()
    assert np.array_equal(occ, orc.occluded_rays(nodes, rays, fill=filled(np.dtype(np.int32), len(rays))))


def test_reference_kernels_default_build_within_tolerance(engine, orc):
    """The kernels as RadeonRays builds them by default (native_recip, OpenCL's 2.5-ulp division): same hits, t within
    1e-4 (BASELINE.json's tolerance) of the exact-division build the oracle restates."""
    flat = FlatScene(SCENES["terrain"]())
    opts = make_options(orc, sample_count=16, sampler=capi.SAMPLER_COSINE, seed=2, bvh_width=2, bvh_quantised=0)
    nodes, surfels = library_tree_and_surfels(flat, opts)
    rays = orc.generate_rays(surfels, None, opts, flat.material_array[0].width)
    with rk.Module("intersect_bvh2_lds_native.co") as mod:
        got = rk.intersect(mod, nodes, rays, filled(orc.INTERSECTION, len(rays)))
    want = orc.intersect_rays(nodes, rays, fill=filled(orc.INTERSECTION, len(rays)))
    same = got["shape_id"] == want["shape_id"]
    # an approximate reciprocal may flip a ray that grazes an edge: allowed, but rare
    assert same.mean() > 0.9995, f"{(~same).sum()} of {len(rays)} rays changed sides"
    both = same & (want["shape_id"] != -1)
    assert both.any()
    t0, t1 = got["uvwt"][both, 3], want["uvwt"][both, 3]
    assert np.abs(t0 - t1).max() <= 1e-4 * max(1.0, float(np.abs(t1).max()))
    # AO from the default build's records vs the oracle's: within the contract
    m = flat.material_array[0]
    a = orc.process_intersections(surfels, rays, got, opts, m.width, m.height)
    b = orc.process_intersections(surfels, rays, want, opts, m.width, m.height)
    wa, wb = a["weight_bits"].view(np.int32), b["weight_bits"].view(np.int32)
    assert np.array_equal(wa, wb)
    cov = wa > 0
    ao_a = a["total_bits"].view(np.int32)[cov] / wa[cov]
    ao_b = b["total_bits"].view(np.int32)[cov] / wb[cov]
    assert np.abs(ao_a - ao_b).mean() <= 1e-4


@pytest.mark.parametrize("name", ["cube_inward", "sphere", "two_section"])
def test_reference_kernels_ray_mask_is_hidden_from_self(engine, orc, name):
    """RR_RAY_MASK (intersect_bvh2_lds.cl:163-166): a ray whose mask is its own mesh id skips that mesh — what
    AOB_MESH_HIDDEN_FROM_SELF does in k_trace's MASKED instantiation."""
    flat = FlatScene(SCENES[name]())
    flags = [capi.MESH_HIDDEN_FROM_SELF] + [0] * (flat.desc.mesh_count - 1)
    for i, f in enumerate(flags):
        flat.mesh_array[i].flags = f
    opts = make_options(orc, sample_count=16, sampler=capi.SAMPLER_COSINE, seed=4, use_any_hit=0, falloff_distance=0.5,
                        accumulate=capi.ACCUMULATE_QUANTISED, bvh_width=2, bvh_quantised=0)
    m = flat.material_array[0]
    with Baker(flat, opts) as b:
        b.build_bvh()
        b.rasterize(0)
        b.trace(0)
        got = b.results(0)
        nodes, surfels = b.nodes(), b.surfels(0)
    _, sidx, smesh = orc.rasterize_ex(flat.desc, 0, opts)
    starts = [i for i in range(len(sidx)) if sidx[i] == 0] + [len(sidx)]
    want = np.zeros(m.width * m.height, dtype=capi.result_dtype())
    unmasked_hits = 0
    with rk.Module("intersect_bvh2_lds_mask_exact.co") as mod:
        for sb, se in zip(starts[:-1], starts[1:]):
            mesh = int(smesh[sb])
            mask = mesh + 1 if flags[mesh] else 0          # shape ids start at 1 (radeon_rays.h:238); 0 masks nothing
            rays = orc.generate_rays(surfels, sidx, opts, m.width, mask=mask, begin=sb, end=se)
            hits = rk.intersect(mod, nodes, rays, filled(orc.INTERSECTION, len(rays)))
            assert np.array_equal(hits["shape_id"], orc.intersect_rays(nodes, rays, use_mask=True, fill=filled(orc.INTERSECTION, len(rays)))["shape_id"])
            if mask:
                assert not (hits["shape_id"] == mask).any()
                rays["extra"][:, 0] = 0
                unmasked_hits += int((rk.intersect(mod, nodes, rays, filled(orc.INTERSECTION, len(rays)))["shape_id"] == mask).sum())
            orc.process_intersections(surfels[sb:se], rays, hits, opts, m.width, m.height, results=want)
    assert bits(got).tobytes() == bits(want).tobytes()
    if name != "two_section":
        assert unmasked_hits > 0           # the mask had something to hide


def ramp_scene(g, jitter=0.0, seed=7):
    """A g x g vertex grid whose height is c * (i + j): the two triangles of a cell get boxes one step apart in y, cells are
    apart in x and z, so all 2 (g-1)^2 primitives have DISTINCT 30-bit Morton codes (g <= 512) — the case in which the
    reference's 30-bit order and the build's 63-bit order must agree primitive for primitive.  jitter: every vertex moved
    by up to that fraction of a cell in x and z (a quarter of it in y), seeded — an irregular mesh that still has that
    property at g = 160, jitter = 0.2 (checked in the test, never assumed)."""
    t = np.linspace(0.0, 1.0, g)
    x, z = np.meshgrid(t, t, indexing="xy")
    i, j = np.meshgrid(np.arange(g), np.arange(g), indexing="xy")
    y = 0.25 * (i + j) / (2.0 * (g - 1))
    if jitter:
        rng = np.random.default_rng(seed)
        h = jitter / (g - 1)
        x = x + rng.uniform(-h, h, x.shape)
        z = z + rng.uniform(-h, h, z.shape)
        y = y + rng.uniform(-h, h, y.shape) * 0.25
    p = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    n = np.tile(np.array([0, 1, 0], dtype=np.float32), (g * g, 1))
    uv = np.stack([x, z], axis=-1).reshape(-1, 2).astype(np.float32)
    return scenes._scene_with([(p, n, uv, scenes._grid_indices(g, g))], 32)


HLBVH_SCENES = {
    "cube": lambda: scenes.cube_scene(32),                                  # 12 triangles, 6 codes: the index fallback
    "ramp12": lambda: ramp_scene(12),                                       # 242 triangles: one 256-work-item workgroup
    "ramp40": lambda: ramp_scene(40),                                       # 3 042 triangles, 48 workgroups
    "jitter160": lambda: ramp_scene(160, jitter=0.2),                       # 50 562 triangles, 198 workgroups, irregular
}


@pytest.mark.parametrize("name", ["cube", "ramp12", "ramp40", "jitter160"])
def test_reference_kernels_hlbvh_build_matches_library_and_oracle(engine, orc, name):
    """build_hlbvh.cl on the GPU — calculate_morton_code_main, (host sort), emit_hierarchy_main, refit_bounds_main —
    against `bvh_builder = 0, morton_adaptive = 0` of the library and the oracle.  The reference sorts 30-bit codes, the
    build 63-bit ones whose top 30 bits are the reference's: where primitives that share a 30-bit code also share the
    63-bit one (then both sides fall back to the index) the two orders, hence the two trees, must be the same tree."""
    flat = FlatScene(HLBVH_SCENES[name]())
    tris = orc.scene_triangles(flat.desc)
    n = len(tris)
    codes63, boxes, scene_box = orc.morton_codes(tris, adaptive=0)
    with rk.Module("build_hlbvh.co") as mod:
        codes30 = rk.morton_codes(mod, boxes, scene_box).astype(np.int64)
        assert np.array_equal(codes30, (codes63 >> np.uint64(33)).astype(np.int64))       # a2: the Morton code itself
        # Two triangles of one quad have the same box, hence the same code at any width: both sides order them by index and
        # split them by index (build_hlbvh.cl:131-150).  What must not happen is two primitives that share the 30-bit code
        # but not the 63-bit one: the reference would split them by index, the build by the finer bits.
        # The scenes here are made so that this cannot happen; a change that breaks that must fail, not skip.
        assert len(np.unique(codes30)) == len(np.unique(codes63)), f"{name}: 63-bit codes collide in their top 30 bits"
        if name != "cube":
            assert len(np.unique(codes30)) == n
        order = np.argsort(codes30, kind="stable")
        one_group = n <= 256            # the code object's max_flat_workgroup_size (OpenCL default)
        # refit_bounds_main has no fence between a child's box store and its sibling's read of it (build_hlbvh.cl:283-312):
        # across workgroups that is a race on this chip, inside one workgroup (one CU, one L1) it is not
        if one_group:
            bufs = _hierarchy_one_group(mod, codes30[order], boxes, order)
        else:
            bufs = rk.hierarchy(mod, codes30[order], boxes, order)
    hl, bb = bufs
    ref_nodes, _ = orc.bvh_build(tris, 0, 0)
    with Baker(flat, make_options(orc, bvh_builder=0, morton_adaptive=0, bvh_width=2, bvh_quantised=0)) as b:
        b.build_bvh()
        lib_nodes = b.nodes()
    assert bits(lib_nodes).tobytes() == bits(ref_nodes).tobytes()
    inner = np.arange(n - 1)
    # topology: children of every inner node, leaves in the sorted order
    assert np.array_equal(hl["left"][inner].astype(np.uint32), lib_nodes["addr_left"][inner])
    assert np.array_equal(hl["right"][inner].astype(np.uint32), lib_nodes["addr_right"][inner])
    assert np.array_equal(hl["left"][n - 1:], order.astype(np.int32))
    leaves = lib_nodes[n - 1:]
    assert np.array_equal(leaves["lmin_or_v0"], tris["v"][order, 0]) and np.array_equal(leaves["rmin_or_v2"], tris["v"][order, 2])
    assert np.array_equal(leaves["prim_id"], tris["prim_id"][order]) and np.array_equal(leaves["shape_id"], tris["shape_id"][order])
    if one_group:
        # boxes: an inner node stores its children's boxes (intersect_bvh2_lds.cl:40-47)
        l, r = hl["left"][inner], hl["right"][inner]
        assert np.array_equal(bb["pmin"][l, :3], lib_nodes["lmin_or_v0"][inner]) and np.array_equal(bb["pmax"][l, :3], lib_nodes["lmax_or_v1"][inner])
        assert np.array_equal(bb["pmin"][r, :3], lib_nodes["rmin_or_v2"][inner]) and np.array_equal(bb["pmax"][r, :3], lib_nodes["rmax"][inner])


def _hierarchy_one_group(mod, sorted_codes, prim_boxes, sorted_indices):
    n = len(sorted_codes)
    bufs = [rk.DeviceBuffer(np.ascontiguousarray(sorted_codes, dtype=np.int32)), rk.DeviceBuffer(prim_boxes),
            rk.DeviceBuffer(np.ascontiguousarray(sorted_indices, dtype=np.int32)),
            rk.DeviceBuffer(nbytes=16 * (2 * n - 1), fill=0xff), rk.DeviceBuffer(nbytes=32 * (2 * n - 1), fill=0),
            rk.DeviceBuffer(nbytes=4 * (2 * n - 1), fill=0)]
    try:
        mod.launch("emit_hierarchy_main", n, [bufs[0], bufs[1], bufs[2], C.c_int32(n), bufs[3], bufs[4]], block=256)
        mod.launch("refit_bounds_main", n, [bufs[4], C.c_int32(n), bufs[3], bufs[5]], block=256)
        return bufs[3].read(rk.HLBVH_NODE, 2 * n - 1), bufs[4].read(rk.BBOX, 2 * n - 1)
    finally:
        for b in bufs:
            b.free()


def test_reference_kernels_two_level_instancing_matches_oracle(engine, orc):
    """The reference stack's instancing path, intersect_bvh2level_skiplinks.cl :134-307 — a top-level hierarchy over
    shapes, each with a world-to-object transform and a shared bottom-level tree, the ray transformed on entry (:105-131,
    :246-253) — run on the GPU on a 13-object scene (12 transformed copies of one rock sharing one tree + the ground):
    the closest hit of every ray equals the oracle's two-level walk (orc_two_level_intersect: the walk DESIGN.md §8's
    flat-versus-two-level measurement is made with) in shape, primitive and t, bit for bit — and the flat walk over the
    flattened scene up to the rounding of object-space arithmetic."""
    sc = scenes.instanced_scene(instances=12, rock_n=20, atlas=128, seed=5)
    flat = FlatScene(sc)
    opts = make_options(orc, sample_count=16, sampler=capi.SAMPLER_COSINE, seed=5, normal_fp16=capi.FP16_OFF)
    flat_nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    surfels, _ = orc.rasterize(flat.desc, 0, opts)
    rays = orc.generate_rays(surfels[::5], None, opts, 128, t_max=1.25)
    tlas, instances, blas = orc.two_level_scene(sc)
    nodes, vertices, faces, shapes, root = rk.two_level_buffers(tlas, instances, blas)
    assert nodes.dtype.itemsize == 32 and shapes.dtype.itemsize == 112 and faces.dtype.itemsize == 20
    with rk.Module("intersect_bvh2level_skiplinks_exact.co") as mod:
        got = rk.intersect_two_level(mod, nodes, vertices, faces, shapes, root, rays, filled(orc.INTERSECTION, len(rays)))
    want, top, bottom = orc.two_level_intersect(tlas, instances, blas, rays, any_hit=False)
    active = rays["extra"][:, 1] != 0
    assert active.all() and top.hits > 0
    assert np.array_equal(got["shape_id"], want["shape_id"])
    hit = want["shape_id"] != -1
    assert np.array_equal(got["prim_id"][hit], want["prim_id"][hit])
    assert bits(got["uvwt"][hit, 3]).tobytes() == bits(want["uvwt"][hit, 3]).tobytes()
    # ... and the flat tree the library walks finds the same surfaces
    fh, _ = orc.flat_intersect(flat_nodes, rays, False)
    same = (fh["shape_id"] != -1) == hit
    assert same.mean() > 0.998
    both = same & hit & (fh["shape_id"] == want["shape_id"])
    assert both.sum() > 0.995 * hit.sum()
    assert np.abs(fh["uvwt"][both, 3] - want["uvwt"][both, 3]).max() < 1e-4
