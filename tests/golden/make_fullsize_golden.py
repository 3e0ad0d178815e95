#!/usr/bin/env python3
"""tests/golden/make_fullsize_golden.py [--configs 2,3] — the COMPLETE images of BASELINE configs[2] (999 698-triangle terrain, 2048^2, 256 spp) and
configs[3] (the same terrain at 4096^2) baked by the CPU oracle, every texel (until round 6 the constants bench.py carries for these
two atlases had been produced by the GPU path, the oracle vouching for a subset of the texels only).

Writes tests/golden/fullsize_crc.json — per config the CRC-32 of the resolved float atlas (what bench.py prints as config.image_crc32 and
tests/test_gpu_fullsize.py asserts), the CRC-32 of the result cells, the ray / hit counts — and tests/golden/fullsize_cells_<config>.npz: the result
cells of every 16th texel in x and y (1/256 of them), so that the GPU box can compare cells without running the oracle.

The oracle's bake of configs[2] takes ~1.5 minutes on 8 cores, configs[3] ~6.  bench.CONFIG2_IMAGE_CRC32 / CONFIG3_IMAGE_CRC32 read the JSON.
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from aobaker_amd import capi, scenes            # noqa: E402
from aobaker_amd.scene import FlatScene         # noqa: E402
from oracle import oracle as orc                # noqa: E402

CONFIGS = {"2": dict(atlas=2048, label="BASELINE configs[2]: 999 698-triangle terrain (grid 708, seed 3), 2048^2, 256 spp, cosine sampler seed 3, any hit, float sums, R32F"),
           "3": dict(atlas=4096, label="BASELINE configs[3]: the same terrain at 4096^2, 256 spp")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    args = ap.parse_args()
    path = os.path.join(HERE, "fullsize_crc.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for c in args.configs.split(","):
        atlas = CONFIGS[c]["atlas"]
        scene = scenes.terrain_scene(grid=708, atlas=atlas, seed=3, bpc=4)
        flat = FlatScene(scene)
        o = orc.default_options()
        o.sample_count, o.sampler, o.accumulate, o.normal_fp16, o.seed, o.use_any_hit = 256, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, 3, 1
        o.bvh_builder = 1                     # (an any-hit answer does not depend on the tree: tests/test_gpu_fullsize.py bakes builders 1 and 2 against this)
        t0 = time.time()
        tris = orc.scene_triangles(flat.desc)
        nodes, _ = orc.bvh_build(tris, 1, 1)
        surfels, sidx = orc.rasterize(flat.desc, 0, o)
        cells, stats = orc.trace(nodes, surfels, sidx, o, atlas, atlas, None, threads=args.threads)
        img = orc.resolve(cells, atlas, atlas, o, channels=1, bpc=4)
        dt = time.time() - t0
        keep = np.zeros((atlas, atlas), dtype=bool); keep[::16, ::16] = True
        idx = np.nonzero(keep.reshape(-1))[0].astype(np.int64)
        np.savez_compressed(os.path.join(HERE, f"fullsize_cells_{c}.npz"), cell_index=idx, total_bits=cells["total_bits"][idx], weight_bits=cells["weight_bits"][idx])
        out[c] = {"config": CONFIGS[c]["label"], "image_crc32": int(zlib.crc32(np.ascontiguousarray(img).tobytes())),
                  "cells_crc32": int(zlib.crc32(np.ascontiguousarray(cells).tobytes())), "surfels": int(len(surfels)), "rays": int(stats.rays), "hits": int(stats.hits),
                  "oracle_seconds": round(dt, 1), "oracle_threads": args.threads, "made_by": "tests/golden/make_fullsize_golden.py (oracle/aob_oracle.c: orc.trace + orc.resolve, every texel)"}
        with open(path, "w") as f:
            json.dump(out, f, indent=1); f.write("\n")
        print(c, out[c], flush=True)


if __name__ == "__main__":
    main()
