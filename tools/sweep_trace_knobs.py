#!/usr/bin/env python3
"""tools/sweep_trace_knobs.py — k_trace time on BASELINE configs[2] over AobOptions::trace_* (the schedule of the fused
trace kernel): how many lanes may wait at a leaf / without a ray before the wavefront runs the leaf / refill section.
The evidence behind the defaults (24 / 16) and behind DESIGN.md §3.1's reading of the leaf section: parking more lanes makes
the leaf section denser and everything else sparser.  One JSON line per setting; product code only (no oracle).

  python tools/sweep_trace_knobs.py [--atlas 2048] [--spp 256] > gpurun_out/knobs.jsonl
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--atlas", type=int, default=2048)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch  # noqa: F401  (one HIP runtime)
    from aobaker_amd import BakeEngine, Baker, capi, scenes
    from aobaker_amd.scene import FlatScene

    BakeEngine.Init(0)
    flat = FlatScene(scenes.terrain_scene(grid=args.grid, atlas=args.atlas, seed=3, bpc=4))

    def run(**knobs):
        o = capi.default_options()
        o.sample_count, o.sampler, o.seed, o.use_any_hit, o.accumulate, o.normal_fp16 = args.spp, capi.SAMPLER_COSINE, 3, 1, capi.ACCUMULATE_FLOAT, capi.FP16_OFF
        for k, v in knobs.items():
            setattr(o, k, v)
        with Baker(flat, o) as b:
            b.build_bvh()
            b.rasterize(0)
            ms = []
            for _ in range(args.reps + 1):
                b.clear_results(0)
                b.trace(0)
                b.sync()
                ms.append(b.last_trace_ms())
            n = b.surfel_count(0)
        best = min(ms[1:])
        print(json.dumps(dict(knobs, trace_ms=round(best, 3), grays_per_s=round(n * args.spp / best / 1e6, 3))), flush=True)

    run()
    for leaf_wait in (1, 4, 8, 12, 16, 20, 24, 32, 48):
        run(trace_leaf_wait=leaf_wait)
    for refill in (4, 8, 16, 20, 24, 28, 32, 40, 48):
        run(trace_refill_idle=refill)
    for parts, part_min in ((1, 16), (2, 16), (4, 16), (8, 16), (8, 8)):
        run(trace_parts=parts, trace_part_min=part_min)
    for fine in (2, 4, 8, 16, 1000000):
        run(trace_fine_div=fine)
    run(trace_bands=2)
    # the two thresholds together (they trade against each other: both decide how many lanes walk in the inner section)
    for refill in (12, 16, 20, 24, 32):
        for leaf_wait in (8, 12, 16, 20):
            run(trace_refill_idle=refill, trace_leaf_wait=leaf_wait)
    BakeEngine.Dispose()


if __name__ == "__main__":
    main()
