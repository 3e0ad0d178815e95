"""GPU parity: every stage of the HIP path against the CPU oracle, through the C ABI.

Bit-exact for integer/index work (coverage, surfel order, BVH topology, quantised sums, u8 image)
and — because both sides follow one arithmetic contract — expected bit-exact for the
float work too; the contractual tolerance for AO values is 1e-4 (BASELINE.json north_star) and is
what the asserts enforce, with the number of non-identical texels reported.
"""
import numpy as np
import pytest

from aobaker_amd import Baker, capi, scenes
from aobaker_amd.scene import FlatScene

pytestmark = pytest.mark.gpu

AO_TOL = 1e-4


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def make_options(orc, **kw):
    o = orc.default_options()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


SCENES = {
    "cube": lambda: scenes.cube_scene(64),
    "cube_inward": lambda: scenes.cube_scene(64, inward=True),
    "plane": lambda: scenes.plane_scene(32),
    "two_section": lambda: scenes.two_section_scene(32),
    "floor_wall": lambda: scenes.floor_wall_scene(32),
    "sphere": lambda: scenes.sphere_scene(n=48, atlas=128),
    "terrain": lambda: scenes.terrain_scene(grid=96, atlas=128),
}


@pytest.mark.parametrize("adaptive", [0, 1])
@pytest.mark.parametrize("builder", [0, 1, 2])
@pytest.mark.parametrize("name", ["cube", "two_section", "sphere", "terrain", "floor_wall", "plane"])
def test_bvh_nodes_bit_exact(engine, orc, name, builder, adaptive):
    """builder 0: LBVH (build_hlbvh.cl); builder 1: PLOC; builder 2: binned SAH top-down over PLOC's clusters (these scenes have
    fewer triangles than clustering stops at: the whole tree is the SAH tree); adaptive: the Morton bit order.  Topology, boxes,
    leaf order, depth."""
    flat = FlatScene(SCENES[name]())
    ref_nodes, ref_depth = orc.bvh_build(orc.scene_triangles(flat.desc), builder, adaptive)
    with Baker(flat, make_options(orc, bvh_builder=builder, morton_adaptive=adaptive)) as b:
        b.build_bvh()
        nodes = b.nodes()
        st = b.stats()
    assert len(nodes) == len(ref_nodes)
    assert st.bvh_depth == ref_depth
    assert st.triangle_count == flat.triangle_count()
    assert bits(nodes).tobytes() == bits(ref_nodes).tobytes()


@pytest.mark.parametrize("name", ["cube", "two_section", "sphere", "terrain", "floor_wall", "plane"])
def test_bvh4_nodes_bit_exact(engine, orc, name):
    """The 4-ary collapse (AobBvh4Node[]) and the Morton-ordered leaf triangles (AobLeafTri[])."""
    flat = FlatScene(SCENES[name]())
    ref2, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    ref4, ref_tris = orc.bvh4_build(ref2)
    with Baker(flat, make_options(orc)) as b:
        b.build_bvh()
        got4, got_tris, got4q = b.nodes4(), b.leaf_tris(), b.nodes4q()
    assert len(got4) == len(ref4) and len(got_tris) == len(ref_tris)
    assert bits(got_tris).tobytes() == bits(ref_tris).tobytes()
    assert bits(got4).tobytes() == bits(ref4).tobytes()
    # and their quantised form (AobBvh4QNode) on the AobBvh2QNode grid
    _, frame = orc.bvh2q_build(ref2)
    assert np.array_equal(got4q, orc.bvh4q_build(ref4, frame))
    # packed: the live AobBvh4Node records (slot 0 holds a ref) in index order, refs renumbered
    live = got4["lo"][:, 0, 3].view(np.uint32) != 0
    assert len(got4q) == int(live.sum())
    got4 = got4[live]
    packed = np.cumsum(live) - 1
    w = got4q.reshape(-1, 4, 4)
    refs = w[:, :, 3]
    refs4 = got4["lo"][:, :, 3].view(np.uint32)
    inner = (refs4 != capi.INVALID_ADDR) & (refs4 < 0x80000000)
    used = refs4 != capi.INVALID_ADDR
    leaf = used & ~inner
    assert np.array_equal(refs[inner], packed[refs4[inner]]) and np.array_equal(refs[leaf], refs4[leaf])
    # a plane is stored as 0x8000 | q (the bytes of the float 1 + q/32768); the decoded boxes contain the fp32 boxes
    assert ((w[:, :, :3][used] & 0x80008000) == 0x80008000).all()
    assert (w[:, :, :3][~used] == 0x8000ffff).all()          # empty slots: the inverted box no ray enters ...
    assert (refs[~used] == 0x80000000).all()                 # ... and a ref that is harmless to follow (leaf triangle 0)
    f64 = frame.astype(np.float64)
    lo = (f64[:3] + (w[:, :, :3] & 0x7fff).astype(np.float64) * f64[3:]).astype(np.float32)
    hi = (f64[:3] + ((w[:, :, :3] >> 16) & 0x7fff).astype(np.float64) * f64[3:]).astype(np.float32)
    assert (lo[used] <= got4["lo"][:, :, :3][used]).all() and (hi[used] >= got4["hi"][:, :, :3][used]).all()


@pytest.mark.parametrize("name", ["cube", "two_section", "sphere", "terrain", "floor_wall", "plane"])
def test_quantised_bvh2_nodes_bit_exact(engine, orc, name):
    """AobBvh2QNode[]: 15-bit child boxes on the scene grid, rounded outward, plus the frame."""
    flat = FlatScene(SCENES[name]())
    ref2, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    refq, ref_frame = orc.bvh2q_build(ref2)
    with Baker(flat, make_options(orc)) as b:
        b.build_bvh()
        gotq, frame = b.nodes2q()
    assert bits(frame).tobytes() == bits(ref_frame).tobytes()
    assert gotq.shape == refq.shape and np.array_equal(gotq, refq)
    # decoded boxes contain the fp32 boxes
    if len(refq):
        q = np.stack([refq[:, 0] & 0xffff, refq[:, 1] & 0xffff, refq[:, 2] & 0xffff, refq[:, 0] >> 16, refq[:, 1] >> 16, refq[:, 2] >> 16], axis=1).astype(np.float64)
        # decode = fmaf(q, scale, origin): exact in float64, then one rounding to float32
        lo = (frame[:3].astype(np.float64) + q[:, :3] * frame[3:].astype(np.float64)).astype(np.float32)
        hi = (frame[:3].astype(np.float64) + q[:, 3:] * frame[3:].astype(np.float64)).astype(np.float32)
        inner = ref2[:len(refq)]
        assert (lo <= inner["lmin_or_v0"]).all() and (hi >= inner["lmax_or_v1"]).all()


@pytest.mark.parametrize("fp16", [capi.FP16_OFF, capi.FP16_RTZ, capi.FP16_RNE])
@pytest.mark.parametrize("name", ["cube", "two_section", "sphere", "terrain"])
def test_surfels_bit_exact(engine, orc, name, fp16):
    flat = FlatScene(SCENES[name]())
    opt = make_options(orc, normal_fp16=fp16)
    ref, ref_idx = orc.rasterize(flat.desc, 0, opt)
    with Baker(flat, opt) as b:
        b.rasterize(0)
        got = b.surfels(0)
    assert len(got) == len(ref) > 0
    assert bits(got).tobytes() == bits(ref).tobytes()


_soup_scene = scenes.soup_scene


@pytest.mark.parametrize("fp16", [capi.FP16_OFF, capi.FP16_RNE])
@pytest.mark.parametrize("name", ["cube", "two_section", "sphere", "terrain", "soup", "soup_large", "sphere_large"])
def test_conservative_surfels_bit_exact(engine, orc, name, fp16):
    """AOB_RASTER_CONSERVATIVE: texels whose square a triangle overlaps get a surfel at the triangle's nearest point."""
    make = {"soup": _soup_scene, "soup_large": lambda: _soup_scene(seed=3, n=20000, atlas=512),
            "sphere_large": lambda: scenes.sphere_scene(n=187, atlas=1024, seed=2)}.get(name) or SCENES[name]
    flat = FlatScene(make())
    opt = make_options(orc, normal_fp16=fp16, raster_mode=capi.RASTER_CONSERVATIVE)
    ref, _ = orc.rasterize(flat.desc, 0, opt)
    plain, _ = orc.rasterize(flat.desc, 0, make_options(orc, normal_fp16=fp16))
    with Baker(flat, opt) as b:
        b.rasterize(0)
        got = b.surfels(0)
    assert len(got) == len(ref) >= len(plain) > 0
    assert bits(got).tobytes() == bits(ref).tobytes()
    if name == "soup":
        assert len(ref) > 1.2 * len(plain)                 # the slivers matter here


@pytest.mark.parametrize("name", ["two_section", "sphere", "soup", "soup_large"])
def test_all_layers_surfels_bit_exact(engine, orc, name):
    """AOB_LAYERS_ALL: every layer of the stencil peel as a section of its own, surfel for surfel."""
    make = {"soup": lambda: _soup_scene(seed=21, n=300, atlas=64, spread=0.08, slivers=False),
            "soup_large": lambda: _soup_scene(seed=4, n=30000, atlas=256, spread=0.03, slivers=False)}.get(name) or SCENES[name]
    flat = FlatScene(make())
    opt = make_options(orc, raster_layers=capi.LAYERS_ALL)
    ref, ref_idx = orc.rasterize(flat.desc, 0, opt)
    plain, _ = orc.rasterize(flat.desc, 0, make_options(orc))
    with Baker(flat, opt) as b:
        b.rasterize(0)
        got = b.surfels(0)
        with pytest.raises(capi.AobError):
            Baker(flat, make_options(orc, raster_layers=capi.LAYERS_ALL, raster_mode=capi.RASTER_CONSERVATIVE))
    assert len(got) == len(ref) >= len(plain) > 0
    assert bits(got).tobytes() == bits(ref).tobytes()
    if name.startswith("soup"):
        assert len(ref) > 1.5 * len(plain)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_all_layers_bake_equals_oracle(engine, orc, mode):
    flat = FlatScene(_soup_scene(seed=9, n=300, atlas=64, spread=0.08, slivers=False))
    opt = make_options(orc, sample_count=8, raster_layers=capi.LAYERS_ALL, **MODES[mode])
    m = flat.material_array[0]
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    want, _ = orc.trace(nodes, surfels, sidx, opt, m.width, m.height, orc.compat_table(opt))
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0)
        got = b.results(0).copy()
        stitched = np.zeros_like(got)
        if MODES[mode]["sampler"] == capi.SAMPLER_COSINE:                  # texel-keyed samples: strips give the same cells
            from aobaker_amd import dist as aod
            for r in range(2):
                y0, y1 = aod.row_window(m.height, r, 2)
                b.rasterize(0, (y0, y1)); b.trace(0, 0, b.surfel_count(0))
                stitched[y0 * m.width:y1 * m.width] = b.results(0)[y0 * m.width:y1 * m.width]
            assert np.array_equal(stitched["total_bits"], got["total_bits"]) and np.array_equal(stitched["weight_bits"], got["weight_bits"])
    assert np.array_equal(got["total_bits"], want["total_bits"]) and np.array_equal(got["weight_bits"], want["weight_bits"])
    if MODES[mode]["sampler"] == capi.SAMPLER_COSINE and MODES[mode]["accumulate"] == capi.ACCUMULATE_FLOAT:
        assert got["weight_bits"].view(np.float32).max() >= 3 * 8          # some texel has three layers or more


@pytest.mark.parametrize("mode", [0, 2])
def test_conservative_bake_equals_oracle_and_stitches(engine, orc, mode):
    """A whole bake in conservative mode == oracle; strips of texel rows stitch to it (a triangle straddling two strips
    gives each strip its texels)."""
    from aobaker_amd import dist as aod
    flat = FlatScene(_soup_scene(seed=5, n=200, atlas=64))
    opt = make_options(orc, sample_count=8, raster_mode=capi.RASTER_CONSERVATIVE, **MODES[mode])
    m = flat.material_array[0]
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    want, _ = orc.trace(nodes, surfels, sidx, opt, m.width, m.height, orc.compat_table(opt))
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0)
        full = b.results(0).copy()
        assert np.array_equal(full["total_bits"], want["total_bits"]) and np.array_equal(full["weight_bits"], want["weight_bits"])
        if mode == 2:                                       # the texel-keyed sampler: strips give the cells of the whole bake
            stitched = np.zeros_like(full)
            for r in range(3):
                y0, y1 = aod.row_window(m.height, r, 3)
                b.rasterize(0, (y0, y1)); b.trace(0, 0, b.surfel_count(0))
                stitched[y0 * m.width:y1 * m.width] = b.results(0)[y0 * m.width:y1 * m.width]
            assert np.array_equal(stitched["total_bits"], full["total_bits"]) and np.array_equal(stitched["weight_bits"], full["weight_bits"])


MODES = [
    dict(sampler=capi.SAMPLER_REFERENCE_COMPAT, accumulate=capi.ACCUMULATE_QUANTISED, use_any_hit=1),
    dict(sampler=capi.SAMPLER_REFERENCE_COMPAT, accumulate=capi.ACCUMULATE_QUANTISED, use_any_hit=0),
    dict(sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, use_any_hit=1, seed=7),
    dict(sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, use_any_hit=0, seed=7,
         clamp_tmax_to_falloff=1),
    dict(sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_QUANTISED, use_any_hit=1, seed=3),
    dict(sampler=capi.SAMPLER_REFERENCE_COMPAT, accumulate=capi.ACCUMULATE_FLOAT, use_any_hit=0),
]


@pytest.mark.parametrize("width", [2, "2q", "4q"])
@pytest.mark.parametrize("mode", range(len(MODES)))
@pytest.mark.parametrize("name", ["cube", "two_section", "sphere", "terrain", "cube_inward"])
def test_trace_results_match_oracle(engine, orc, name, mode, width):
    """4q: the quantised 4-ary nodes (the default); 2 / 2q: the reference's 64-byte BVH2 nodes / their 32-byte quantised form.
    Each against the oracle walking the same structure.  (The quad-per-ray walk of the fp32 4-ary nodes was retired in round 5.)"""
    flat = FlatScene(SCENES[name]())
    opt = make_options(orc, sample_count=24, bvh_width=int(str(width)[0]), bvh_quantised=int(str(width).endswith("q")), **MODES[mode])
    m = flat.material_array[0]
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    hemi = orc.compat_table(opt) if opt.sampler == capi.SAMPLER_REFERENCE_COMPAT else None
    ref, stats = orc.trace(nodes, surfels, sidx, opt, m.width, m.height, hemi, threads=4)
    with Baker(flat, opt) as b:
        b.build_bvh()
        b.rasterize(0)
        b.trace(0)
        got = b.results(0)
    if opt.accumulate == capi.ACCUMULATE_QUANTISED:
        assert np.array_equal(got["total_bits"], ref["total_bits"])
        assert np.array_equal(got["weight_bits"], ref["weight_bits"])
    else:
        gt, gw = got["total_bits"].view(np.float32), got["weight_bits"].view(np.float32)
        rt, rw = ref["total_bits"].view(np.float32), ref["weight_bits"].view(np.float32)
        assert np.array_equal(gw != 0, rw != 0)
        cov = rw != 0
        err = np.abs(gt[cov] / gw[cov] - rt[cov] / rw[cov])
        assert err.max() <= AO_TOL
        # the arithmetic contract makes the sums themselves identical
        assert np.array_equal(got["total_bits"], ref["total_bits"])
        assert np.array_equal(got["weight_bits"], ref["weight_bits"])
    assert 0 < stats.rays <= len(surfels) * opt.sample_count     # surfels with a NaN frame cast nothing


@pytest.mark.parametrize("bleed_mode", [capi.BLEED_REFERENCE, capi.BLEED_NEAREST])
@pytest.mark.parametrize("fmt", [(1, 1), (4, 1), (1, 2), (2, 2), (1, 4), (3, 4), (4, 4)])
def test_resolve_formats(engine, orc, fmt, bleed_mode):
    channels, bpc = fmt
    sc = scenes.cube_scene(64, channels=channels, bpc=bpc)          # six charts with gutters: the bleed has work to do
    flat = FlatScene(sc)
    opt = make_options(orc, sample_count=16, bleed_mode=bleed_mode, bleed_distance=5 if bleed_mode == capi.BLEED_REFERENCE else 3)
    m = flat.material_array[0]
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0); b.resolve(0)
        res = b.results(0)
        img = np.zeros(m.width * m.height * channels * bpc, dtype=np.uint8)
        b.download(0, img)
    ref = orc.resolve(res, m.width, m.height, opt, channels, bpc)
    assert np.array_equal(img, ref)


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("name", ["cube", "two_section", "terrain", "plane", "cube_inward"])
def test_bake_end_to_end(engine, orc, name, mode):
    """BakeEngine.Bake(scene) against orc_bake: the final texture bytes."""
    opt = make_options(orc, sample_count=16, **MODES[mode])
    bpc = 1 if opt.accumulate == capi.ACCUMULATE_QUANTISED else 4
    mk = {"cube": lambda: scenes.cube_scene(64, bpc=bpc), "two_section": lambda: scenes.two_section_scene(32, bpc=bpc),
          "terrain": lambda: scenes.terrain_scene(grid=96, atlas=128, bpc=bpc), "plane": lambda: scenes.plane_scene(32, bpc=bpc),
          "cube_inward": lambda: scenes.cube_scene(64, inward=True, bpc=bpc)}[name]
    gpu_scene, cpu_scene = mk(), mk()
    engine.Options = opt
    engine.Bake(gpu_scene)
    orc.bake(FlatScene(cpu_scene).desc, opt, threads=4)
    g = gpu_scene.GetMaterial(0).GetAOTexture()
    c = cpu_scene.GetMaterial(0).GetAOTexture()
    err = np.abs(g.ao() - c.ao()).max()
    assert err <= AO_TOL
    assert np.array_equal(g.GetData(), c.GetData())
    if name == "plane":
        assert (g.ao() == 1.0).all()            # nothing above an open plane
    if name == "cube_inward" and bpc == 1:
        assert (g.GetData()[g.GetData() != 0].size == 0)   # a closed cube is black inside


def test_bake_texture_overload(engine, orc):
    """Bake(Scene*, BakeTexture*): only the named material's texture is written."""
    from aobaker_amd.scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices
    sc = Scene()
    a = sc.AddMaterial(MaterialObject("A", 32, 32))
    b_ = sc.AddMaterial(MaterialObject("B", 16, 16))
    q1 = scenes.quad((0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0))
    q2 = scenes.quad((0, 0.3, 0), (1, 0.3, 0), (1, 0.3, 1), (0, 0.3, 1), (0, 1, 0))
    o = sc.AddObject(BakeObject("O"))
    o.AddSection(Section("a", RenderMesh(make_vertices(*q1[:3]), q1[3]), a))
    o.AddSection(Section("b", RenderMesh(make_vertices(*q2[:3]), q2[3]), b_))
    engine.Options = make_options(orc, sample_count=8)
    engine.Bake(sc, b_.GetAOTexture())
    assert (a.GetAOTexture().GetData() == 0xFF).all()          # untouched (BakeTexture.cpp:63)
    assert (b_.GetAOTexture().GetData() == 255).all()          # top quad sees the sky
    engine.Bake(sc, a.GetAOTexture())
    assert a.GetAOTexture().ao()[16, 16] < 0.35                # the bottom quad sits under the top one
    assert (b_.GetAOTexture().GetData() == 255).all()


def test_empty_and_degenerate_inputs(engine, orc):
    from aobaker_amd.scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices
    opt = make_options(orc, sample_count=4)
    # no meshes at all: every material is written black (BakeEngine.cpp:404-709 with no sections)
    sc = Scene(); sc.AddMaterial(MaterialObject("A", 8, 8))
    engine.Options = opt
    engine.Bake(sc)
    assert (sc.GetMaterial(0).GetAOTexture().GetData() == 0).all()
    # degenerate triangles only (the reference's import bug feeds (0,0,0) triangles, BakeObject.cpp:96)
    p = np.zeros((3, 3), np.float32); n = np.tile(np.float32([0, 1, 0]), (3, 1)); uv = np.zeros((3, 2), np.float32)
    sc = Scene(); m = sc.AddMaterial(MaterialObject("A", 8, 8))
    o = sc.AddObject(BakeObject("O")); o.AddSection(Section("s", RenderMesh(make_vertices(p, n, uv), np.uint32([0, 0, 0, 0, 1, 2])), m))
    engine.Bake(sc)
    assert (m.GetAOTexture().GetData() == 0).all()
    # zero normals (LoadFromCube's fixture, BakeObject.cpp:197-207): NaN frame, contributes nothing
    q = scenes.quad((0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 0, 0))
    sc = Scene(); m = sc.AddMaterial(MaterialObject("A", 8, 8))
    o = sc.AddObject(BakeObject("O")); o.AddSection(Section("s", RenderMesh(make_vertices(*q[:3]), q[3]), m))
    engine.Bake(sc)
    cpu = Scene(); mc = cpu.AddMaterial(MaterialObject("A", 8, 8))
    oc = cpu.AddObject(BakeObject("O")); oc.AddSection(Section("s", RenderMesh(make_vertices(*q[:3]), q[3]), mc))
    orc.bake(FlatScene(cpu).desc, opt)
    assert np.array_equal(m.GetAOTexture().GetData(), mc.GetAOTexture().GetData())


@pytest.mark.parametrize("size", [(1, 1), (3, 200), (37, 53), (100, 100), (257, 9), (64, 7)])
@pytest.mark.parametrize("mode", [0, 2, 3])
def test_ragged_atlas_sizes(engine, orc, size, mode):
    """Atlases that are no multiple of the 8x8 trace tile, of the 64-surfel wavefront or of the 2048-texel scan tile
    (the reference's BakeTexture takes any size, BakeTexture.cpp:52-67): whole bake == oracle, strips stitch."""
    from aobaker_amd import dist as aod
    mk = lambda: scenes.terrain_scene(grid=24, atlas=size, bpc=1 if MODES[mode]["accumulate"] == capi.ACCUMULATE_QUANTISED else 4)
    opt = make_options(orc, sample_count=5, **MODES[mode])
    gpu, cpu = mk(), mk()
    engine.Options = opt
    engine.Bake(gpu)
    orc.bake(FlatScene(cpu).desc, opt, threads=4)
    assert np.array_equal(gpu.GetMaterial(0).GetAOTexture().GetData(), cpu.GetMaterial(0).GetAOTexture().GetData())
    if mode == 2 and size[1] > 8:
        flat = FlatScene(mk())
        with Baker(flat, opt) as b:
            b.build_bvh(); b.rasterize(0); b.trace(0)
            full = b.results(0).copy()
            stitched = np.zeros_like(full)
            for r in range(3):
                y0, y1 = aod.row_window(size[1], r, 3)
                b.rasterize(0, (y0, y1)); b.trace(0, 0, b.surfel_count(0))
                stitched[y0 * size[0]:y1 * size[0]] = b.results(0)[y0 * size[0]:y1 * size[0]]
        assert np.array_equal(stitched["total_bits"], full["total_bits"]) and np.array_equal(stitched["weight_bits"], full["weight_bits"])


@pytest.mark.parametrize("chunk,spp", [(4096, 8), (1000, 7), (64, 64), (0xFFFFFF, 300)])
def test_chunk_local_sample_indexing(engine, orc, chunk, spp):
    """The REFERENCE_COMPAT sampler reads Hemisphere[k * S + i] with k the surfel's index WITHIN its chunk of
    min(rayChunkSize / S, 65535) texels (BakeEngine.cpp:359-364,598-628): small chunks wrap many times over a section."""
    mk = lambda: scenes.terrain_scene(grid=24, atlas=(96, 80))
    opt = make_options(orc, sample_count=spp, ray_chunk_size=chunk, **MODES[0])
    gpu, cpu = mk(), mk()
    engine.Options = opt
    engine.Bake(gpu)
    orc.bake(FlatScene(cpu).desc, opt, threads=4)
    assert np.array_equal(gpu.GetMaterial(0).GetAOTexture().GetData(), cpu.GetMaterial(0).GetAOTexture().GetData())
    flat = FlatScene(mk())
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    want, _ = orc.trace(nodes, surfels, sidx, opt, 96, 80, orc.compat_table(opt))
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0)
        got = b.results(0)
    assert np.array_equal(got["total_bits"], want["total_bits"]) and np.array_equal(got["weight_bits"], want["weight_bits"])
    assert len(surfels) > 4 * max(1, min(chunk // spp, 65535)) or chunk == 0xFFFFFF        # the index really wraps


@pytest.mark.parametrize("spp", [1, 2, 31, 33, 127, 1000, 4097])
@pytest.mark.parametrize("mode", [0, 2, 3, 4])
def test_odd_and_large_sample_counts(engine, orc, spp, mode):
    """Sample counts that do not divide into the kernel's four parts per tile, below the 32 samples a part needs, and far
    above the reference's 100: whole bake == oracle."""
    mk = lambda: scenes.two_section_scene(16, bpc=1 if MODES[mode]["accumulate"] == capi.ACCUMULATE_QUANTISED else 4)
    opt = make_options(orc, sample_count=spp, **MODES[mode])
    gpu, cpu = mk(), mk()
    engine.Options = opt
    engine.Bake(gpu)
    st = orc.bake(FlatScene(cpu).desc, opt, threads=8)
    assert engine.LastStats.ray_count == st.rays and st.rays % spp == 0
    for i in range(gpu.GetMaterialCount()):
        assert np.array_equal(gpu.GetMaterial(i).GetAOTexture().GetData(), cpu.GetMaterial(i).GetAOTexture().GetData())


@pytest.mark.parametrize("structure", [(2, 0, 0), (4, 1, 1), (2, 1, 1)])
@pytest.mark.parametrize("bad", ["nan", "inf", "huge", "tiny", "far"])
def test_hostile_coordinates(engine, orc, bad, structure):
    """A NaN or infinite vertex, coordinates of 1e30 / 1e-30, a mesh 1e8 away from the origin: builders, quantisation and
    the walk end, and agree with the oracle (such triangles are never hit, `common.cl:193-216`)."""
    from aobaker_amd.scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices

    def mk():
        p, n, uv, idx = scenes.terrain_mesh(grid=36, seed=1)            # 2 592 triangles: more than one PLOC tile
        p = p.copy()
        if bad == "nan":
            p[10] = np.nan; p[700, 2] = np.nan
        elif bad == "inf":
            p[10, 1] = np.inf; p[900, 0] = -np.inf
        elif bad == "huge":
            p *= np.float32(1e30)
        elif bad == "tiny":
            p *= np.float32(1e-30)
        else:
            p += np.float32(1e8)
        sc = Scene(); m = sc.AddMaterial(MaterialObject("A", 48, 48))
        o = sc.AddObject(BakeObject("O")); o.AddSection(Section("s", RenderMesh(make_vertices(p, n, uv), idx), m))
        return sc
    opt = make_options(orc, sample_count=6, **MODES[0])
    opt.bvh_width, opt.bvh_quantised, opt.bvh_builder = structure
    gpu, cpu = mk(), mk()
    engine.Options = opt
    engine.Bake(gpu)
    st = orc.bake(FlatScene(cpu).desc, opt, threads=8)
    # AobStats::ray_count is covered surfels x spp; the oracle counts the rays it cast, and casts none from a
    # surfel whose position or normal is not finite
    assert engine.LastStats.ray_count >= st.rays and (bad in ("nan", "inf") or engine.LastStats.ray_count == st.rays)
    assert np.array_equal(gpu.GetMaterial(0).GetAOTexture().GetData(), cpu.GetMaterial(0).GetAOTexture().GetData())


@pytest.mark.parametrize("structure", [(4, 1, 2), (4, 1, 1), (2, 1, 1), (2, 0, 0)])
@pytest.mark.parametrize("shape", ["point", "line", "flat_no_bias", "flat_stack"])
@pytest.mark.parametrize("mode", [0, 2, 3])
def test_scenes_without_extent(engine, orc, mode, shape, structure):
    """Scene boxes with no extent: every vertex at one point (all triangles degenerate, the quantisation frame's scale 0 on
    all three axes — the empty child slots of the quantised 4-ary nodes cannot be told from boxes and lead to leaf triangle
    0, bvh_build.hip k_emit_nodes4q), on one line, in one plane with ray_bias 0, and two coincident planes that do occlude
    each other's rays at bias 0.  Texel charts and normals are those of a quad, so rays ARE cast; the walk ends and the
    texture is the oracle's."""
    from aobaker_amd.scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices
    uv = np.float32([[0.05, 0.05], [0.95, 0.05], [0.95, 0.95], [0.05, 0.95]])
    nrm = np.tile(np.float32([0, 1, 0]), (4, 1))
    quad = np.float32([[0, 0, 0], [1, 0, 0], [1, 0, 1], [0, 0, 1]])
    p = {"point": np.tile(np.float32([0.3, 0.4, 0.5]), (4, 1)),
         "line": quad * np.float32([1, 0, 0]) + np.float32([0, 0.25, 0.5]),
         "flat_no_bias": quad + np.float32([0, 0.125, 0]),
         "flat_stack": quad}[shape]
    idx = np.uint32([0, 1, 2, 0, 2, 3] * (3 if shape == "flat_stack" else 1))

    def mk(bpc):
        sc = Scene(); m = sc.AddMaterial(MaterialObject("A", 24, 24, 1, bpc))
        o = sc.AddObject(BakeObject("O")); o.AddSection(Section("s", RenderMesh(make_vertices(p, nrm, uv), idx), m))
        return sc
    opt = make_options(orc, sample_count=16, ray_bias=0.0, **MODES[mode])
    opt.bvh_width, opt.bvh_quantised, opt.bvh_builder = structure
    bpc = 1 if opt.accumulate == capi.ACCUMULATE_QUANTISED else 4
    gpu, cpu = mk(bpc), mk(bpc)
    engine.Options = opt
    engine.Bake(gpu)
    st = orc.bake(FlatScene(cpu).desc, opt, threads=4)
    assert st.rays > 0 and engine.LastStats.ray_count == st.rays
    assert np.array_equal(gpu.GetMaterial(0).GetAOTexture().GetData(), cpu.GetMaterial(0).GetAOTexture().GetData())


@pytest.mark.parametrize("spp", [128, 200])
@pytest.mark.parametrize("mode", [0, 2, 4])
def test_fine_units_at_the_end_of_the_queues(engine, orc, mode, spp):
    """Enough tiles per XCD queue (>= 8) and enough samples (>= 128) for the last eighth of every queue to be cut into twice as
    many sample parts: the cells are the oracle's whatever the cut."""
    flat = FlatScene(scenes.terrain_scene(grid=40, atlas=(96, 72)))              # 6 912 surfels: 108 tiles, 14 per queue
    opt = make_options(orc, sample_count=spp, **MODES[mode])
    m = flat.material_array[0]
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    want, _ = orc.trace(nodes, surfels, sidx, opt, m.width, m.height, orc.compat_table(opt), threads=8)
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0)
        got = b.results(0)
    assert np.array_equal(got["total_bits"], want["total_bits"]) and np.array_equal(got["weight_bits"], want["weight_bits"])


@pytest.mark.parametrize("mode", [0, 2, 4])
def test_trace_schedule_knobs_do_not_change_results(engine, orc, mode):
    """AobOptions::trace_*  move work between wavefronts; the result cells stay bit-identical."""
    flat = FlatScene(scenes.terrain_scene(grid=40, atlas=(128, 72)))

    def cells(**knobs):
        with Baker(flat, make_options(orc, sample_count=160, **MODES[mode], **knobs)) as b:
            b.build_bvh(); b.rasterize(0); b.trace(0)
            return b.results(0).copy()

    want = cells()
    for knobs in (dict(trace_parts=1), dict(trace_parts=8, trace_part_min=4), dict(trace_fine_div=2), dict(trace_fine_div=1000000),
                  dict(trace_refill_idle=1, trace_leaf_wait=1), dict(trace_refill_idle=64, trace_leaf_wait=64), dict(trace_bands=2),
                  dict(trace_refill_idle=0, trace_leaf_wait=0, trace_parts=0, trace_part_min=0, trace_fine_div=0, trace_bands=0),
                  dict(trace_cut=1), dict(trace_cut=4), dict(trace_cut=19), dict(trace_cut=32, trace_refill_idle=42, trace_parts=1)):
        got = cells(**knobs)
        assert np.array_equal(got["total_bits"], want["total_bits"]) and np.array_equal(got["weight_bits"], want["weight_bits"]), knobs
    with pytest.raises(capi.AobError):
        cells(trace_refill_idle=65)
    with pytest.raises(capi.AobError):
        cells(trace_cut=3)


def test_invalid_arguments_are_reported(engine, orc):
    flat = FlatScene(scenes.plane_scene(8))
    opt = make_options(orc, sample_count=0)
    with pytest.raises(capi.AobError) as e:
        Baker(flat, opt)
    assert e.value.code == -1 and "sample_count" in str(e.value)
    opt = make_options(orc)
    with Baker(flat, opt) as b:
        with pytest.raises(capi.AobError):
            b.trace(0)                      # BVH not built yet
        with pytest.raises(capi.AobError):
            b.rasterize(3)                  # no such material
        # aob_bake_add_results: another handle's cells on the same device are added word by word (what an RCCL sum-reduce does
        # between devices); a cell count that is not the material's is refused
        import ctypes as C
        b.build_bvh(); b.rasterize(0); b.trace(0)
        once = b.results(0).copy()
        with Baker(flat, opt) as other:
            other.build_bvh(); other.rasterize(0); other.trace(0)
            other.sync()                    # the sum runs on b's stream: the other handle's work must have finished
            ptr, nbytes = C.c_void_p(), C.c_uint64()
            capi.check(other.lib.aob_bake_results_device_ptr(other.handle, 0, C.byref(ptr), C.byref(nbytes)))
            cells = nbytes.value // 8
            assert cells == len(once)
            assert b.lib.aob_bake_add_results(b.handle, 0, ptr, cells - 1) == -1 and b"cells" in b.lib.aob_last_error()
            capi.check(b.lib.aob_bake_add_results(b.handle, 0, ptr, cells))
            b.sync()
            twice = b.results(0)
            # aob_bake_add_results_from: the other HANDLE named — the call waits for its stream itself (no sync by the caller:
            # the trace below may still be running when the sum is enqueued)
            other.clear_results(0); other.trace(0)
            capi.check(b.lib.aob_bake_add_results_from(b.handle, 0, other.handle, 0))
            b.sync()
            thrice = b.results(0)
            assert b.lib.aob_bake_add_results_from(b.handle, 0, other.handle, 7) == -1
        assert np.array_equal(twice["total_bits"].view(np.int32), 2 * once["total_bits"].view(np.int32))
        assert np.array_equal(twice["weight_bits"].view(np.int32), 2 * once["weight_bits"].view(np.int32))
        assert np.array_equal(thrice["total_bits"].view(np.int32), 3 * once["total_bits"].view(np.int32))
        assert np.array_equal(thrice["weight_bits"].view(np.int32), 3 * once["weight_bits"].view(np.int32))


def test_handles_work_from_threads_that_never_called_init(engine, orc):
    """aob_init on the main thread, aob_bake_create and every stage from a worker thread: the worker creates its handle on
    the process's device (the most recent aob_init of any thread), and a handle made on one thread is driven from another —
    each entry point selects the handle's device itself (api.hip check_bake).  Same cells as the main thread's bake."""
    import threading
    flat = FlatScene(SCENES["sphere"]())
    opt = make_options(orc, sample_count=16, **MODES[2])
    with Baker(flat, opt) as main_handle:
        main_handle.build_bvh(); main_handle.rasterize(0); main_handle.trace(0)
        want = main_handle.results(0).copy()
        out = {}

        def worker():
            try:
                with Baker(flat, opt) as b:                  # created on a thread without an aob_init of its own
                    b.build_bvh(); b.rasterize(0); b.trace(0)
                    out["own"] = b.results(0).copy()
                main_handle.clear_results(0); main_handle.trace(0)      # a handle of another thread
                out["foreign"] = main_handle.results(0).copy()
            except Exception as e:                           # noqa: BLE001 - reported by the assert below
                out["error"] = repr(e)
        t = threading.Thread(target=worker); t.start(); t.join()
    assert "error" not in out, out.get("error")
    assert bits(out["own"]).tobytes() == bits(want).tobytes() and bits(out["foreign"]).tobytes() == bits(want).tobytes()


@pytest.mark.parametrize("name,parts", [("terrain", 3), ("cube", 2), ("sphere", 5), ("two_section", 2), ("instanced", 4)])
def test_sharded_trace_equals_full_trace(engine, orc, name, parts):
    """aob_bake_shard_bounds + aob_bake_trace(begin, end): the multi-GPU decomposition on one GPU.
    Parts are traced one after the other into the same result buffer; single-section materials get
    tile-row aligned parts that own disjoint strips of cells.  (`instanced`: ranges of a list whose 13 sections go in one launch.)"""
    flat = FlatScene(scenes.instanced_scene(instances=12, rock_n=20, atlas=256, seed=5, bpc=4) if name == "instanced" else SCENES[name]())
    opt = make_options(orc, sample_count=16, sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, seed=11)
    m = flat.material_array[0]
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0)
        b.trace(0)
        full = b.results(0)
        n = b.surfel_count(0)
        bounds, strips, overlapping = b.shard_bounds(0, parts)
        assert bounds[0] == 0 and bounds[-1] == n and all(bounds[i] <= bounds[i + 1] for i in range(parts))
        # (`overlapping`: the parts' cells are not one strip each — several sections; whether two sections share a TEXEL is another matter)
        assert overlapping == (name in ("two_section", "instanced"))
        b.clear_results(0)
        owned = np.zeros(m.width * m.height, dtype=np.int32)
        for r in range(parts):
            b.clear_results(0)
            b.trace(0, bounds[r], bounds[r + 1])
            part = b.results(0)
            touched = (part["weight_bits"] != 0)
            if name == "instanced":
                # sections that share no texel: every surfel has its own cell, so a part's cells are the full bake's, or untouched
                assert np.array_equal(part[touched], full[touched])
            if not overlapping and bounds[r + 1] > bounds[r]:
                c0, c1 = strips[r]
                assert c0 % (8 * m.width) == 0                       # starts on a tile row
                assert not touched[:c0].any() and not touched[c1 + 1:].any()
                assert np.array_equal(part[c0:c1 + 1], full[c0:c1 + 1])
            owned += touched
        if not overlapping or name == "instanced":
            assert owned.max() == 1 and np.array_equal(owned > 0, full["weight_bits"] != 0)
        sizes = [bounds[i + 1] - bounds[i] for i in range(parts)]
        if name in ("terrain", "sphere"):
            assert max(sizes) - min(sizes) <= 8 * m.width        # balanced to within one tile row


@pytest.mark.parametrize("mode", [3, 0])
def test_instanced_scene_with_transforms(engine, orc, mode):
    """BASELINE cfg5 in miniature: many objects with their own yaw/pitch/roll + non-uniform scale +
    translation (world = [p 1] * model, normals through the model matrix itself —
    RasterizeVertexShader.hlsl:30-32), one shared material, falloff scoring with the ray clamped to
    the falloff distance (mode 3) or the reference's defaults (mode 0)."""
    mk = lambda bpc: scenes.instanced_scene(instances=12, rock_n=20, atlas=256, seed=5, bpc=bpc)
    opt = make_options(orc, sample_count=16, **MODES[mode])
    bpc = 1 if opt.accumulate == capi.ACCUMULATE_QUANTISED else 4
    gpu, cpu = mk(bpc), mk(bpc)
    flat = FlatScene(gpu)
    assert flat.desc.mesh_count == 13
    ref_nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc))
    with Baker(flat, opt) as b:
        b.build_bvh()
        assert bits(b.nodes()).tobytes() == bits(ref_nodes).tobytes()      # 13 shapes, 13 transforms
        b.rasterize(0)
        ref_surfels, _ = orc.rasterize(flat.desc, 0, opt)
        got = b.surfels(0)
        assert len(got) == len(ref_surfels) and bits(got).tobytes() == bits(ref_surfels).tobytes()
    engine.Options = opt
    engine.Bake(gpu)
    orc.bake(FlatScene(cpu).desc, opt, threads=8)
    g, c = gpu.GetMaterial(0).GetAOTexture(), cpu.GetMaterial(0).GetAOTexture()
    assert np.array_equal(g.GetData(), c.GetData())
    ao = g.ao()
    assert ao.max() > 0.9 and 0.02 < (ao < 0.9).mean() < 0.97         # rocks darken the ground and each other; gutters are black


@pytest.mark.parametrize("mode", [2, 3, 0])
@pytest.mark.parametrize("name", ["terrain", "sphere", "cube_inward"])
@pytest.mark.parametrize("builder", [1, 2])
def test_ploc_builder_end_to_end(engine, orc, name, mode, builder):
    """bvh_builder = 1 (and 2: the SAH top) through every derived structure and the trace, against the oracle's twin."""
    flat = FlatScene(SCENES[name]())
    opt = make_options(orc, sample_count=16, bvh_builder=builder, **MODES[mode])
    m = flat.material_array[0]
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc), builder)
    surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    hemi = orc.compat_table(opt) if opt.sampler == capi.SAMPLER_REFERENCE_COMPAT else None
    ref, st1 = orc.trace(nodes, surfels, sidx, opt, m.width, m.height, hemi, threads=8)
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0)
        got = b.results(0)
        ref4, _ = orc.bvh4_build(nodes)
        assert bits(b.nodes4()).tobytes() == bits(ref4).tobytes()
    assert np.array_equal(got["total_bits"], ref["total_bits"]) and np.array_equal(got["weight_bits"], ref["weight_bits"])
    # same image as the LBVH.  Node visits: on these SMALL scenes (18 050 / 4 608 triangles) the two builders are level once
    # both sort by the adaptive Morton order — PLOC 3 015 317 vs LBVH 3 002 547 inner visits on this terrain, +0.4 % —
    # so the bound is "not worse by more than 2 %"; PLOC's 5 % advantage is a property of the 1 M-triangle
    # tree and is asserted there (tests/test_gpu_fullsize.py::test_config3_ploc_visits_fewer_nodes_than_lbvh)
    opt0 = make_options(orc, sample_count=16, bvh_builder=0, **MODES[mode])
    nodes0, _ = orc.bvh_build(orc.scene_triangles(flat.desc), 0)
    ref0, st0 = orc.trace(nodes0, surfels, sidx, opt0, m.width, m.height, hemi, threads=8)
    assert np.array_equal(ref0["weight_bits"], ref["weight_bits"])
    if opt.use_any_hit:
        assert np.array_equal(ref0["total_bits"], ref["total_bits"])
    if name != "cube_inward":
        assert st1.inner_visits < 1.02 * st0.inner_visits
        # ... and that order is what helps a flat scene: the strict x, y, z interleave visits more nodes
        if name == "terrain":
            nodes_xyz, _ = orc.bvh_build(orc.scene_triangles(flat.desc), 1, 0)
            _, st_xyz = orc.trace(nodes_xyz, surfels, sidx, opt, m.width, m.height, hemi, threads=8)
            assert st1.inner_visits < st_xyz.inner_visits


# ---- section visibility and material range ---------------------------------------------------
def _flagged_scene(bpc):
    """Two materials, four sections with every flag combination, a range on one material."""
    from aobaker_amd.scene import BakeObject, MaterialObject, RenderMesh, Scene, Section, make_vertices
    sc = Scene()
    ma = sc.AddMaterial(MaterialObject("a", 48, 48, 1, bpc))
    mb = sc.AddMaterial(MaterialObject("b", 32, 32, 1, bpc))
    mb.SetAOMaxRange(0.6)

    def part(quadspec, mat, vis_self, vis_others, obj):
        v, n, uv, idx = quadspec
        sec = obj.AddSection(Section("s", RenderMesh(make_vertices(v, n, uv), idx), mat))
        sec.SetVisibleToSelf(vis_self); sec.SetVisibleToOthers(vis_others)

    o1 = sc.AddObject(BakeObject("floor"))
    part(scenes.quad((0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0), (0, 0), (1, 0.5)), ma, True, True, o1)
    o2 = sc.AddObject(BakeObject("walls", rotation=(0.0, 0.3, 0.0)))
    part(scenes.quad((0.2, 0, 0.2), (0.8, 0, 0.2), (0.8, 0.7, 0.2), (0.2, 0.7, 0.2), (0, 0, 1), (0, 0.5), (0.5, 1)), ma, False, True, o2)
    part(scenes.quad((0.2, 0, 0.8), (0.2, 0.7, 0.8), (0.8, 0.7, 0.8), (0.8, 0, 0.8), (0, 0, -1), (0.5, 0.5), (1, 1)), ma, True, False, o2)
    o3 = sc.AddObject(BakeObject("lid"))
    part(scenes.quad((0, 0.5, 0), (0, 0.5, 1), (1, 0.5, 1), (1, 0.5, 0), (0, -1, 0)), mb, False, False, o3)
    return sc


@pytest.mark.parametrize("structure", [(4, 1), (2, 1), (2, 0)])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_section_visibility_and_material_range(engine, orc, mode, structure):
    opt = make_options(orc, sample_count=16, material_ranges=1, bvh_width=structure[0], bvh_quantised=structure[1], **MODES[mode])
    bpc = 1 if opt.accumulate == capi.ACCUMULATE_QUANTISED else 4
    gpu_scene, cpu_scene, plain = _flagged_scene(bpc), _flagged_scene(bpc), _flagged_scene(bpc)
    engine.Options = opt
    engine.Bake(gpu_scene)
    orc.bake(FlatScene(cpu_scene).desc, opt, threads=4)
    for i in range(2):
        assert np.array_equal(gpu_scene.GetMaterial(i).GetAOTexture().GetData(), cpu_scene.GetMaterial(i).GetAOTexture().GetData())
    # and the flags do something: without them the bake differs
    for o in range(plain.GetObjectCount()):
        for sidx in range(plain.GetObject(o).GetSectionCount()):
            plain.GetObject(o).GetSection(sidx).SetVisibleToSelf(True); plain.GetObject(o).GetSection(sidx).SetVisibleToOthers(True)
    engine.Bake(plain)
    assert not np.array_equal(plain.GetMaterial(0).GetAOTexture().GetData(), gpu_scene.GetMaterial(0).GetAOTexture().GetData())


def test_the_retired_quad_per_ray_structure_is_refused(engine, orc):
    """bvh_width 4 with bvh_quantised 0 selected tools/experiments/trace4.hip until round 4; the library no longer carries it and says so."""
    opt = make_options(orc, sample_count=16, bvh_width=4, bvh_quantised=0)
    engine.Options = opt
    with pytest.raises(capi.AobError) as e:
        engine.Bake(_flagged_scene(1))
    assert e.value.code == -5 and "retired" in str(e.value)      # AOB_ERR_UNSUPPORTED


# ---- deep traversal stacks: entries beyond the 8 / 11 kept in LDS go to the HBM array -----------------------------
def _hairball_scene(bpc, n=3000, seed=11):
    """A floor patch under a ball of long thin triangles in random directions: their boxes all overlap, so a ray
    defers many children at once (the oracle's walk of the same tree reaches stack depths far beyond 11)."""
    from aobaker_amd.scene import BakeObject, RenderMesh, Section, make_vertices
    rng = np.random.default_rng(seed)
    sc = scenes.plane_scene(24, bpc=bpc)
    c = np.float32([0.5, 0.6, 0.5])
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    w = rng.normal(size=(n, 3)); w -= (w * d).sum(1, keepdims=True) * d; w /= np.linalg.norm(w, axis=1, keepdims=True)
    o = c + rng.normal(scale=0.05, size=(n, 3))
    v = np.stack([o - 0.45 * d, o + 0.45 * d, o + 0.02 * w], axis=1).astype(np.float32).reshape(-1, 3)
    nrm = np.repeat(np.cross(d, w).astype(np.float32), 3, axis=0)
    uv = np.zeros((3 * n, 2), dtype=np.float32)
    obj = sc.AddObject(BakeObject("hairball"))
    obj.AddSection(Section("Geometry", RenderMesh(make_vertices(v, nrm, uv), np.arange(3 * n, dtype=np.uint32)), None))
    return sc


@pytest.mark.parametrize("structure", [(4, 1), (2, 1), (2, 0)])
@pytest.mark.parametrize("mode", [1, 2])
def test_deep_stacks_spill_to_hbm_and_come_back(engine, orc, mode, structure):
    opt = make_options(orc, sample_count=16, bvh_width=structure[0], bvh_quantised=structure[1], **MODES[mode])
    bpc = 1 if opt.accumulate == capi.ACCUMULATE_QUANTISED else 4
    gpu_scene, cpu_scene = _hairball_scene(bpc), _hairball_scene(bpc)
    engine.Options = opt
    engine.Bake(gpu_scene)
    orc.bake(FlatScene(cpu_scene).desc, opt, threads=8)
    g, c = gpu_scene.GetMaterial(0).GetAOTexture(), cpu_scene.GetMaterial(0).GetAOTexture()
    assert np.array_equal(g.GetData(), c.GetData())
    ao = g.ao()
    assert 0.02 < ao.mean() < 0.9          # the ball occludes a good part of the sky, not all of it


# ---- a rank's strip of texel rows (aob_bake_rasterize_rows) ------------------------------------------------------
@pytest.mark.parametrize("parts", [2, 3])
@pytest.mark.parametrize("name", ["cube", "terrain", "two_section", "instanced"])
def test_row_window_bakes_stitch_to_the_full_bake(engine, orc, name, parts):
    """Rasterising and tracing the atlas strip by strip gives the cells of the whole bake (what bench.py does per rank).
    `instanced`: 13 sections of one material that share no texel — one launch for all of them, per strip too; `two_section`:
    sections that do overlap — a launch each."""
    from aobaker_amd import dist as aod
    flat = FlatScene(scenes.instanced_scene(instances=12, rock_n=20, atlas=256, seed=5, bpc=4) if name == "instanced" else SCENES[name]())
    opt = make_options(orc, sample_count=16, **MODES[2])
    m = flat.material_array[0]
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0)
        full = b.results(0).copy()
        n_full = b.surfel_count(0)
        stitched = np.zeros_like(full)
        n_sum = 0
        for r in range(parts):
            y0, y1 = aod.row_window(m.height, r, parts)
            b.rasterize(0, (y0, y1))
            n = b.surfel_count(0)
            n_sum += n
            b.trace(0, 0, n)
            part = b.results(0)
            cells = slice(y0 * m.width, y1 * m.width)
            outside = np.ones(len(part), dtype=bool); outside[cells] = False
            assert not part["weight_bits"][outside].any()           # nothing outside the strip
            stitched[cells] = part[cells]
        with pytest.raises(capi.AobError):
            b.rasterize(0, (8, m.height + 8))
    assert n_sum == n_full
    assert np.array_equal(stitched["total_bits"], full["total_bits"]) and np.array_equal(stitched["weight_bits"], full["weight_bits"])


# ---- tile cuts: the records k_tile_cuts writes against the oracle's orc_tile_cut, and the per-part records against the oracle's rays --
def _slots_of(n4q16, node):
    """The non-empty slots of a quantised 4-ary node as cut rows {ref, x, y, z plane words}."""
    rows = []
    for j in range(4):
        c = n4q16[node, 4 * j:4 * j + 4]
        if c[0] == 0x8000ffff and c[1] == 0x8000ffff and c[2] == 0x8000ffff:
            continue
        rows.append((int(c[3]), int(c[0]), int(c[1]), int(c[2])))
    return rows


def _leaf_slots_reached(orc, n4q, n4q16, frame, lt, rays, rows):
    """The leaf slots ANY of the rays enters below the given slots (the oracle's slab test, level by level; no early exit): what a
    walk that starts at these slots can ever test."""
    reached, frontier = set(), list(rows)
    while frontier:
        nxt = []
        for at in range(0, len(frontier), 64):
            chunk = frontier[at:at + 64]
            _, _, masks = orc.cut_walk(n4q, frame, lt, rays, np.array(chunk, dtype=np.uint32), want_masks=True)
            entered = int(np.bitwise_or.reduce(masks))
            for e, row in enumerate(chunk):
                if not (entered >> e) & 1:
                    continue
                if row[0] & 0x80000000:
                    reached.add(row[0])
                else:
                    nxt += _slots_of(n4q16, row[0])
        frontier = nxt
    return reached


def test_tile_cuts_match_the_oracle_and_every_part_record_reaches_what_the_root_walk_reaches(engine, orc):
    """Without sample parts (the table sampler) the cut of every tile is the oracle's (same breadth-first rule, same budget, bit for
    bit).  With them every part has its own record — the common cut culled by the part's wedge: its entries are slots of the tree,
    none below another, and the leaf slots the part's rays (the oracle's rays, the oracle's slab test, no early exit) can reach
    from the record are EXACTLY those they reach from the root (so the test holds for ANY rule that picks a part's entries)."""
    flat = FlatScene(scenes.terrain_scene(grid=96, atlas=64, seed=3, bpc=4))
    for sampler, spp in ((capi.SAMPLER_COSINE, 64), (capi.SAMPLER_COSINE, 256), (capi.SAMPLER_REFERENCE_COMPAT, 16)):
        opt = make_options(orc, sample_count=spp, bvh_builder=2, **MODES[2])
        opt.sampler = sampler
        if sampler == capi.SAMPLER_REFERENCE_COMPAT:
            opt.accumulate = capi.ACCUMULATE_QUANTISED
        with Baker(flat, opt) as b:
            b.build_bvh(); b.rasterize(0); b.trace(0)
            cuts = b.tile_cuts(0)
            n4q, (_, frame) = b.nodes4q(), b.nodes2q()
            lt = b.leaf_tris()
            surfels = b.surfels(0)
        tiles, records, _ = cuts.shape
        # 8 x 8 tiles; one record per tile, or one per sample part: 4 coarse (+ 8 fine where the queues end in half-length units)
        want_records = 1 if sampler == capi.SAMPLER_REFERENCE_COMPAT else (12 if spp == 256 else 4)
        assert tiles == 64 and records == want_records
        n4q16 = np.ascontiguousarray(n4q).view(np.uint32).reshape(-1, 16)
        slot_planes, parent = {}, {}
        for node in range(len(n4q16)):
            for row in _slots_of(n4q16, node):
                slot_planes[row[0]] = row[1:]
                parent[row[0]] = node
        grid = surfels.reshape(64, 64)
        shrink = np.float32(0.25)                       # k_tile_cuts: the tile's box pulled in by a quarter of its extent on every side
        for t in (range(tiles) if records == 1 else (0, 7, 9, 15, 27, 36, 47, 63)):
            ty, tx = divmod(t, 8)
            sub = np.ascontiguousarray(grid[8 * ty:8 * ty + 8, 8 * tx:8 * tx + 8].reshape(-1))
            pos = sub["position"].astype(np.float32)
            pmin, pmax = pos.min(axis=0), pos.max(axis=0)
            pull = shrink * (pmax - pmin)
            want = orc.tile_cut(n4q, frame, pmin + pull, pmax - pull, 32, 1)
            common = {int(x) for x in want[:, 0]}
            rays = np.ascontiguousarray(orc.generate_rays(sub, None, opt, 64, hemisphere=orc.compat_table(opt) if sampler == capi.SAMPLER_REFERENCE_COMPAT else None))
            rays = rays.reshape(64, spp)
            for r in range(records):
                rec = cuts[t, r]
                refs, planes = rec[:32], rec[32:].reshape(32, 3)
                used = int((planes[:, 0] != 0x8000ffff).sum())
                got = np.concatenate([refs[:used, None], planes[:used]], axis=1)
                assert (planes[used:] == 0x8000ffff).all()
                if records == 1:
                    assert np.array_equal(got, want), (t, r)
                    continue
                # slots of the tree, each once, none an ancestor of another
                rows = [tuple(int(v) for v in x) for x in got.tolist()]
                mine = {row[0] for row in rows}
                assert len(mine) == len(rows), (t, r)
                for row in rows:
                    assert slot_planes[row[0]] == row[1:], (t, r, row)
                    node = parent[row[0]]
                    while node != 0:
                        assert node not in mine, (t, r, row, node)
                        node = parent[node]
                assert mine <= common, (t, r)
                # the part's samples: coarse part r of 4, or fine part r - 4 of 8
                parts, q = (4, r) if r < 4 else (8, r - 4)
                part_rays = np.ascontiguousarray(rays[:, spp * q // parts:spp * (q + 1) // parts].reshape(-1))
                from_root = _leaf_slots_reached(orc, n4q, n4q16, frame, lt, part_rays, _slots_of(n4q16, 0))
                from_record = _leaf_slots_reached(orc, n4q, n4q16, frame, lt, part_rays, rows)
                assert from_record == from_root, (t, r, len(from_root), len(from_record))


# ---- closest-hit rays below tile cuts (round 5) ----------------------------------------------------------------------
def _trace_order(surfels, sidx, width):
    """The order k_trace walks a material's surfels in (raster.hip k_tile_*): per section, by 8x8 texel tile (tiles row-major), row-major
    inside a tile; the sections one after the other.  Returns the permutation of the row-major per-section list."""
    starts = [i for i in range(len(surfels)) if sidx[i] == 0] + [len(surfels)]
    order = []
    for a, b in zip(starts[:-1], starts[1:]):
        x, y = surfels["texel_x"][a:b].astype(np.int64), surfels["texel_y"][a:b].astype(np.int64)
        key = ((y // 8) * ((width + 7) // 8) + x // 8) * 64 + (y % 8) * 8 + x % 8
        order.append(a + np.argsort(key, kind="stable"))
    return np.concatenate(order) if order else np.zeros(0, dtype=np.int64)


def _rocks(atlas=96, bpc=4):
    return scenes.instanced_scene(instances=5, rock_n=20, atlas=atlas, seed=5, bpc=bpc)


@pytest.mark.gpu
@pytest.mark.parametrize("name,scene,kw", [
    ("rocks, falloff clamp", _rocks, dict(clamp_tmax_to_falloff=1, falloff_distance=1.25)),
    ("rocks, short reach", _rocks, dict(clamp_tmax_to_falloff=1, falloff_distance=0.3)),
    ("terrain, unbounded", lambda: scenes.terrain_scene(grid=64, atlas=64, seed=3, bpc=4), dict()),
    ("sphere, falloff clamp", lambda: scenes.sphere_scene(n=40, atlas=96, seed=2, bpc=4), dict(clamp_tmax_to_falloff=1, falloff_distance=0.2)),
])
@pytest.mark.parametrize("sampler", [capi.SAMPLER_COSINE, capi.SAMPLER_REFERENCE_COMPAT])
def test_closest_hit_below_tile_cuts_is_the_oracles_walk_of_the_same_cuts(engine, orc, name, scene, kw, sampler):
    """Closest-hit rays start below their tile's cut, as any-hit rays do (round 5), the entries a ray entered walked deepest first.  Checked
    three ways: (1) every tile's cut is the oracle's (orc_tile_cut_reach: k_tile_cuts' rule, slots beyond the rays' reach dropped) — with
    one record per tile bit for bit, with a record per sample part as a sub-list in the same order; (2) the result cells are those of the
    ORACLE WALKING THE SAME CUTS IN THE SAME ORDER (orc_cut_walk_closest on the tiles of the trace order, scored by
    orc_process_intersections): equal by construction; (3) they are also the cells of the oracle's walk from the root (orc.trace) — the
    minimum over a ray's hits does not depend on the order they are found in."""
    flat = FlatScene(scene())
    spp = 32
    opt = make_options(orc, sample_count=spp, sampler=sampler, accumulate=capi.ACCUMULATE_FLOAT, use_any_hit=0, seed=5, bvh_builder=1, **kw)
    m = flat.material_array[0]
    with Baker(flat, opt) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0); b.sync()
        assert b.stats().trace_cuts == 1
        cuts = b.tile_cuts(0)
        n4q, (_, frame), lt = b.nodes4q(), b.nodes2q(), b.leaf_tris()
        surfels = b.surfels(0)
        got = b.results(0)
    o_surfels, sidx = orc.rasterize(flat.desc, 0, opt)
    assert np.array_equal(surfels, o_surfels)
    order = _trace_order(surfels, sidx, m.width)
    ordered = np.ascontiguousarray(surfels[order])
    tiles, records, _ = cuts.shape
    assert tiles == (len(ordered) + 63) // 64                 # one launch: the sections share no texel
    t_max = np.float32(opt.ray_tmax)
    if opt.clamp_tmax_to_falloff and opt.falloff_distance < opt.ray_tmax:
        t_max = np.float32(opt.falloff_distance)
    reach = float((t_max + np.float32(abs(opt.ray_bias))) * np.float32(1.01))
    hemi = orc.compat_table(opt) if sampler == capi.SAMPLER_REFERENCE_COMPAT else None
    # (the table sampler indexes its table by the surfel's place in its section's row-major list: the rays are made in that order and permuted)
    rays_all = np.ascontiguousarray(orc.generate_rays(surfels, sidx, opt, m.width, hemisphere=hemi, t_max=float(t_max))).reshape(len(surfels), spp)
    rays_all = np.ascontiguousarray(rays_all[order])
    hits = np.zeros((len(ordered), spp), dtype=orc.INTERSECTION)
    shrink = np.float32(0.25)
    dropped = 0
    for t in range(tiles):
        sub = ordered[64 * t:64 * t + 64]
        rays = np.ascontiguousarray(rays_all[64 * t:64 * t + 64].reshape(-1))
        # k_tile_cuts takes the box of the surfels that have a tangent frame: a surfel without one casts no ray (inactive records)
        active = rays.reshape(len(sub), spp)["extra"][:, 0, 1] != 0
        pos = sub["position"].astype(np.float32)[active]
        if len(pos) == 0:
            continue
        pmin, pmax = pos.min(axis=0), pos.max(axis=0)
        pull = shrink * (pmax - pmin)
        want = orc.tile_cut_reach(n4q, frame, pmin + pull, pmax - pull, pmin, pmax, reach, 32)
        unbounded = orc.tile_cut_reach(n4q, frame, pmin + pull, pmax - pull, pmin, pmax, 0.0, 32)
        dropped += int(unbounded.shape != want.shape or not np.array_equal(unbounded, want))
        for r in range(records):
            rec = cuts[t, r]
            refs, planes = rec[:32], rec[32:].reshape(32, 3)
            used = int((planes[:, 0] != 0x8000ffff).sum())
            rows = np.concatenate([refs[:used, None], planes[:used]], axis=1)
            if records == 1:
                assert np.array_equal(rows, want), (name, t)
            else:
                # a part's record: the common cut culled by the part's wedge — a sub-list in the same order
                at = 0
                for row in rows.tolist():
                    while at < len(want) and want[at].tolist() != row:
                        at += 1
                    assert at < len(want), (name, t, r, row)
                    at += 1
        _, tt = orc.cut_walk_closest(n4q, frame, lt, rays, want, order=1)
        _, t_root = orc.walk4q_t(n4q, frame, lt, rays, any_hit=False)
        assert np.array_equal(tt, t_root), (name, t)
        h = hits[64 * t:64 * t + 64].reshape(-1)
        h["shape_id"] = np.where(tt < t_max, 0, -1)
        h["uvwt"][:, 3] = tt
    if "short" in name:
        assert dropped > 0                      # the case is there for the reach: some tile's cut must differ from the unbounded one
    want_cells = orc.process_intersections(ordered, rays_all.reshape(-1), hits.reshape(-1), opt, m.width, m.height, falloff=float(opt.falloff_distance))
    assert np.array_equal(got, want_cells), name
    nodes, _ = orc.bvh_build(orc.scene_triangles(flat.desc), 1)
    ref, _ = orc.trace(nodes, o_surfels, sidx, opt, m.width, m.height, hemi, threads=4)
    assert np.array_equal(got, ref), name


def test_tile_cuts_that_cannot_be_allocated_are_reported_for_every_material(engine, orc):
    """a bake of TWO materials whose tile cuts cannot be allocated walks from the root — the same cells — and says so: AobStats::trace_cuts
    stays 2 whatever the second material's rasterisation does, aob_last_error() holds the warning, and the allocation is not tried launch after launch."""
    import ctypes
    lib = capi.load()
    opt = make_options(orc, sample_count=16, sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, use_any_hit=1, seed=2)
    gpu_scene, cpu_scene = _flagged_scene(4), _flagged_scene(4)
    lib.aob_debug_fail_tile_cuts(1)
    try:
        engine.Options = opt
        engine.Bake(gpu_scene)
        st = engine.LastStats
        assert st.trace_cuts == 2
        assert b"warning:" in (lib.aob_last_error() or b"")
    finally:
        lib.aob_debug_fail_tile_cuts(0)
    orc.bake(FlatScene(cpu_scene).desc, opt, threads=4)
    for i in range(2):
        assert np.array_equal(gpu_scene.GetMaterial(i).GetAOTexture().GetData(), cpu_scene.GetMaterial(i).GetAOTexture().GetData())
    engine.Bake(gpu_scene)                      # and with the memory back the next bake has its cuts again
    assert engine.LastStats.trace_cuts == 1
    engine.Options = None
