#!/usr/bin/env python3
"""Regenerates the committed golden vectors from the CPU oracle.

The reference has no fixtures of its own and cannot run here, so these vectors are
the build's: BASELINE configs[0] — the 12-triangle cube, 256^2 atlas, 16 spp, seed 1 — baked by the
oracle in the reference's mode (REFERENCE_COMPAT sampler, quantised sums, R8 output) and in the
north-star mode (COSINE sampler, float sums, R32F output), plus digests of the intermediate stages;
and the same for a small 7 938-triangle terrain (128^2 atlas) whose AO is not trivially white.
They pin the oracle against regressions, and the GPU path against the oracle at a size the GPU box
can check without running the oracle.

    python tests/golden/make_golden.py      # rewrites tests/golden/{cube_cfg1,terrain_small,features_v2}.npz

features_v2: a 250-triangle soup (slivers, overlapping charts) baked with the options the reference lacks —
conservative rasterisation + nearest-texel dilation, all layers of the stencil peel, a section hidden from itself
under a material range.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from aobaker_amd import capi, scenes            # noqa: E402
from aobaker_amd.scene import FlatScene         # noqa: E402
from oracle import oracle as orc                # noqa: E402


def options(mode):
    o = orc.default_options()
    o.sample_count = 16
    o.seed = 1
    o.bvh_width, o.bvh_quantised, o.bvh_builder, o.morton_adaptive = 2, 0, 0, 0      # the reference stack's own builder (build_hlbvh.cl) and node layout (intersect_bvh2_lds.cl:40-52)
    if mode == "cosine":
        o.sampler, o.accumulate, o.normal_fp16 = capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF
    return o


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    for name, make, size in (("cube_cfg1", lambda bpc: scenes.cube_scene(256, bpc=bpc), 256),
                             ("terrain_small", lambda bpc: scenes.terrain_scene(grid=64, atlas=128, bpc=bpc), 128)):
        out = {}
        for mode, bpc in (("compat", 1), ("cosine", 4)):
            sc = make(bpc)
            flat = FlatScene(sc)
            o = options(mode)
            nodes, depth = orc.bvh_build(orc.scene_triangles(flat.desc), 0, 0)
            surfels, sidx = orc.rasterize(flat.desc, 0, o)
            hemi = orc.compat_table(o) if mode == "compat" else None
            results, stats = orc.trace(nodes, surfels, sidx, o, size, size, hemi, threads=os.cpu_count() or 1)
            image = orc.resolve(results, size, size, o, 1, bpc)
            st = orc.bake(flat.desc, o, threads=os.cpu_count() or 1)
            assert np.array_equal(image, sc.GetMaterial(0).GetAOTexture().GetData())
            out[f"{mode}_image"] = image
            out[f"{mode}_nodes_sha256"] = np.array(digest(nodes))
            out[f"{mode}_surfels_sha256"] = np.array(digest(surfels))
            out[f"{mode}_results_sha256"] = np.array(digest(results))
            out[f"{mode}_counts"] = np.array([len(nodes), depth, len(surfels), st.rays, st.hits], dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        for k, v in out.items():
            print(name, k, v.shape if v.ndim else v)


FEATURES = {
    # options the reference does not have, on a triangle soup with slivers and overlapping charts,
    # with the library's default traversal structure (PLOC, quantised 4-ary nodes, leaf pairs)
    "conservative": (4, dict(sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, normal_fp16=capi.FP16_OFF,
                             raster_mode=capi.RASTER_CONSERVATIVE, bleed_mode=capi.BLEED_NEAREST, bleed_distance=2)),
    "layers": (1, dict(raster_layers=capi.LAYERS_ALL, use_any_hit=0)),
    "masks": (4, dict(sampler=capi.SAMPLER_COSINE, accumulate=capi.ACCUMULATE_FLOAT, material_ranges=1)),
}


def features_scene(bpc):
    sc = scenes.soup_scene(seed=7, n=250, atlas=64, spread=0.06, bpc=bpc)
    sc.GetMaterial(0).SetAOMaxRange(0.35)
    return sc


def features_options(name):
    o = orc.default_options()
    o.sample_count, o.seed = 16, 1
    for k, v in FEATURES[name][1].items():
        setattr(o, k, v)
    return o


def main_features():
    out = {}
    for name, (bpc, _) in FEATURES.items():
        sc = features_scene(bpc)
        if name == "masks":
            sc.GetObject(0).GetSection(0).SetVisibleToSelf(False)
        flat = FlatScene(sc)
        o = features_options(name)
        surfels, _ = orc.rasterize(flat.desc, 0, o)
        st = orc.bake(flat.desc, o, threads=os.cpu_count() or 1)
        out[f"{name}_image"] = sc.GetMaterial(0).GetAOTexture().GetData().copy()
        out[f"{name}_surfels_sha256"] = np.array(digest(surfels))
        out[f"{name}_counts"] = np.array([len(surfels), st.rays, st.hits], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "features_v2.npz"), **out)
    for k, v in out.items():
        print("features_v2", k, v.shape if v.ndim else v)


if __name__ == "__main__":
    main()
    main_features()
