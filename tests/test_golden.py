"""Committed golden vectors (tests/golden/cube_cfg1.npz, made by tests/golden/make_golden.py):
BASELINE configs[0], the reference's own CPU-runnable case — 12-triangle cube, 256^2, 16 spp."""
import hashlib
import os

import numpy as np
import pytest

from aobaker_amd import capi, scenes
from aobaker_amd.scene import FlatScene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = np.load(os.path.join(GOLDEN, "cube_cfg1.npz"))
CASES = {"cube_cfg1": (lambda bpc: scenes.cube_scene(256, bpc=bpc), GOLD),
         "terrain_small": (lambda bpc: scenes.terrain_scene(grid=64, atlas=128, bpc=bpc), np.load(os.path.join(GOLDEN, "terrain_small.npz")))}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def options(orc_or_capi_defaults, mode):
    o = orc_or_capi_defaults
    o.sample_count = 16
    o.seed = 1
    if mode == "cosine":
        o.sampler, o.accumulate, o.normal_fp16 = capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF
    return o


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("mode,bpc", [("compat", 1), ("cosine", 4)])
def test_oracle_reproduces_golden(orc, mode, bpc, case):
    make, GOLD = CASES[case]
    sc = make(bpc)
    flat = FlatScene(sc)
    o = options(orc.default_options(), mode)
    o.bvh_width, o.bvh_quantised, o.bvh_builder, o.morton_adaptive = 2, 0, 0, 0   # the vectors were made with build_hlbvh.cl's LBVH on the reference's node layout
    nodes, depth = orc.bvh_build(orc.scene_triangles(flat.desc), 0, 0)
    surfels, sidx = orc.rasterize(flat.desc, 0, o)
    assert digest(nodes) == str(GOLD[f"{mode}_nodes_sha256"])
    assert digest(surfels) == str(GOLD[f"{mode}_surfels_sha256"])
    st = orc.bake(flat.desc, o, threads=os.cpu_count() or 1)
    assert [len(nodes), depth, len(surfels), st.rays, st.hits] == GOLD[f"{mode}_counts"].tolist()
    assert np.array_equal(sc.GetMaterial(0).GetAOTexture().GetData(), GOLD[f"{mode}_image"])


@pytest.mark.parametrize("structure", [(2, 1, 0), (4, 0, 0), (4, 1, 0), (4, 1, 1), (2, 0, 1)])
@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("mode,bpc", [("compat", 1), ("cosine", 4)])
def test_other_traversal_structures_give_the_golden_image(orc, mode, bpc, case, structure):
    """Every (width, quantised, builder) combination — incl. the GPU default (4, 1, PLOC) — bakes the same image."""
    make, GOLD = CASES[case]
    sc = make(bpc)
    o = options(orc.default_options(), mode)
    o.bvh_width, o.bvh_quantised, o.bvh_builder = structure
    orc.bake(FlatScene(sc).desc, o, threads=os.cpu_count() or 1)
    assert np.array_equal(sc.GetMaterial(0).GetAOTexture().GetData(), GOLD[f"{mode}_image"])


def test_golden_cube_is_plausible():
    """Independent of any implementation: a convex cube baked on its outside sees only sky."""
    img = GOLD["compat_image"].reshape(256, 256)
    assert (img[4:124, 4:80] == 255).all()          # interior of the first chart
    assert img[0, 0] == 255                          # gutter texel bled from the chart 2 texels away
    f = GOLD["cosine_image"].view(np.float32).reshape(256, 256)
    assert (f[4:124, 4:80] == 1.0).all()
    nodes, depth, surfels, rays, hits = GOLD["cosine_counts"].tolist()
    assert nodes == 23 and surfels == 6 * 81 * 124 and rays == surfels * 16 and hits == 0


def test_golden_terrain_has_real_occlusion():
    g = CASES["terrain_small"][1]
    img = g["cosine_image"].view(np.float32)
    assert 0.05 < img.min() < 0.6 and img.max() == 1.0 and 0.7 < img.mean() < 0.99
    nodes, depth, surfels, rays, hits = g["cosine_counts"].tolist()
    assert nodes == 2 * 7938 - 1 and surfels == 128 * 128 and 0 < hits < rays


@pytest.mark.gpu
@pytest.mark.parametrize("structure", ["reference", "default"])
@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("mode,bpc", [("compat", 1), ("cosine", 4)])
def test_gpu_reproduces_golden(engine, mode, bpc, case, structure):
    """The HIP path against the committed vectors — no oracle involved on the GPU box.
    "reference": LBVH + the reference's 64-byte nodes (what the vectors were made with): every stage digest.
    "default": the library's defaults (PLOC, quantised 4-ary nodes): surfels, result cells and image."""
    from aobaker_amd import Baker
    make, GOLD = CASES[case]
    sc = make(bpc)
    flat = FlatScene(sc)
    o = options(capi.default_options(), mode)
    if structure == "reference":
        o.bvh_width, o.bvh_quantised, o.bvh_builder, o.morton_adaptive = 2, 0, 0, 0
    with Baker(flat, o) as b:
        b.build_bvh(); b.rasterize(0); b.trace(0); b.resolve(0); b.download(0)
        if structure == "reference":
            assert digest(b.nodes()) == str(GOLD[f"{mode}_nodes_sha256"])
        assert digest(b.surfels(0)) == str(GOLD[f"{mode}_surfels_sha256"])
        assert digest(b.results(0)) == str(GOLD[f"{mode}_results_sha256"])
        st = b.stats()
    want = GOLD[f"{mode}_counts"].tolist()
    assert [st.node_count, st.surfel_count, st.ray_count] == [want[0], want[2], want[3]]
    if structure == "reference":
        assert st.bvh_depth == want[1]
    assert np.array_equal(sc.GetMaterial(0).GetAOTexture().GetData(), GOLD[f"{mode}_image"])


# ---- features_v2.npz: the options the reference lacks, on a triangle soup (made by the same script) -------------------
FEATURES_GOLD = np.load(os.path.join(GOLDEN, "features_v2.npz"))
FEATURES = {
    "conservative": (4, dict(sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, normal_fp16=capi.FP16_OFF,
                             raster_mode=capi.RASTER_CONSERVATIVE, bleed_mode=capi.BLEED_NEAREST, bleed_distance=2)),
    "layers": (1, dict(raster_layers=capi.LAYERS_ALL, use_any_hit=0)),
    "masks": (4, dict(sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, material_ranges=1)),
}


def features_scene(name):
    sc = scenes.soup_scene(seed=7, n=250, atlas=64, spread=0.06, bpc=FEATURES[name][0])
    sc.GetMaterial(0).SetAOMaxRange(0.35)
    if name == "masks":
        sc.GetObject(0).GetSection(0).SetVisibleToSelf(False)
    return sc


def features_options(o, name):
    o.sample_count, o.seed = 16, 1
    for k, v in FEATURES[name][1].items():
        setattr(o, k, v)
    return o


@pytest.mark.parametrize("name", sorted(FEATURES))
def test_oracle_reproduces_feature_golden(orc, name):
    sc = features_scene(name)
    flat = FlatScene(sc)
    o = features_options(orc.default_options(), name)
    surfels, _ = orc.rasterize(flat.desc, 0, o)
    assert digest(surfels) == str(FEATURES_GOLD[f"{name}_surfels_sha256"])
    st = orc.bake(flat.desc, o, threads=os.cpu_count() or 1)
    assert [len(surfels), st.rays, st.hits] == FEATURES_GOLD[f"{name}_counts"].tolist()
    assert np.array_equal(sc.GetMaterial(0).GetAOTexture().GetData(), FEATURES_GOLD[f"{name}_image"])


def test_feature_goldens_are_plausible():
    """Implementation-independent relations between the vectors: the same soup gives more surfels when every overlapped
    texel / every layer counts; a section hidden from itself under a short range sees (almost) only sky."""
    n_cons, n_layers, n_masks = (int(FEATURES_GOLD[f"{k}_counts"][0]) for k in ("conservative", "layers", "masks"))
    assert n_cons > 1.3 * n_masks and n_layers > 1.05 * n_masks and n_masks > 500
    masks = FEATURES_GOLD["masks_image"].view(np.float32)
    assert FEATURES_GOLD["masks_counts"][2] == 0 and set(np.unique(masks).tolist()) <= {0.0, 1.0}     # no hits: covered texels white
    cons = FEATURES_GOLD["conservative_image"].view(np.float32)
    assert (cons > 0).sum() >= n_cons and 0.0 < cons[cons > 0].min() < 0.9                # real occlusion; dilation only adds texels


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FEATURES))
def test_gpu_reproduces_feature_golden(engine, name):
    """The HIP path against the committed vectors — no oracle involved."""
    from aobaker_amd import Baker
    sc = features_scene(name)
    flat = FlatScene(sc)
    o = features_options(capi.default_options(), name)
    with Baker(flat, o) as b:
        b.rasterize(0)
        assert digest(b.surfels(0)) == str(FEATURES_GOLD[f"{name}_surfels_sha256"])
    engine.Options = o
    engine.Bake(sc)
    assert np.array_equal(sc.GetMaterial(0).GetAOTexture().GetData(), FEATURES_GOLD[f"{name}_image"])
    assert engine.LastStats.ray_count == int(FEATURES_GOLD[f"{name}_counts"][1])


def test_the_headline_images_are_pinned_by_a_complete_oracle_bake():
    """tests/golden/fullsize_crc.json is written by tests/golden/make_fullsize_golden.py — the oracle baking BASELINE configs[2] and configs[3] completely,
    every texel — and bench.py's expected image CRCs ARE its numbers (they used to be constants the GPU path had produced)."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    g = json.load(open(os.path.join(root, "tests", "golden", "fullsize_crc.json")))
    assert g["2"]["image_crc32"] == bench.CONFIG2_IMAGE_CRC32 and g["2"]["rays"] == 2048 * 2048 * 256
    assert g["3"]["image_crc32"] == bench.CONFIG3_IMAGE_CRC32 and g["3"]["rays"] == 4096 * 4096 * 256
    for c, n in (("2", 2048), ("3", 4096)):
        assert "every texel" in g[c]["made_by"] and g[c]["surfels"] == n * n
        sub = __import__("numpy").load(os.path.join(root, "tests", "golden", f"fullsize_cells_{c}.npz"))
        assert len(sub["cell_index"]) == (n // 16) ** 2
