#!/usr/bin/env python3
"""tools/sweep_knobs.py [--mode anyhit|closest] [--reps 3] field=v1,v2,... [field=...] — the trace stage of BASELINE configs[2] under every combination
of the given AobOptions fields (the run-time schedule knobs: trace_refill_idle, trace_leaf_wait, trace_scalar_min, trace_parts, ...), one process, one scene:
best-of-reps HIP-event times of the tile cuts, the horizon tables and k_trace, and the CRC-32 of the result cells (equal CRCs = equal results).
Product code only (no oracle).

  PYTHONPATH=. python tools/sweep_knobs.py trace_refill_idle=8,12,16,24 trace_leaf_wait=12,20,28
"""
import argparse
import itertools
import json
import os
import sys
import zlib

sys.path.insert(0, os.getcwd())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="anyhit")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("fields", nargs="+")
    a = ap.parse_args()
    import numpy as np
    from aobaker_amd import BakeEngine, Baker, capi, scenes
    from aobaker_amd.scene import FlatScene
    BakeEngine.Init(0)
    flat = FlatScene(scenes.terrain_scene(grid=708, atlas=2048, seed=3, bpc=4))
    names = [f.split("=")[0] for f in a.fields]
    values = [[int(v) for v in f.split("=")[1].split(",")] for f in a.fields]
    for combo in itertools.product(*values):
        o = capi.default_options()
        o.sample_count, o.sampler, o.seed, o.accumulate, o.normal_fp16 = a.spp, capi.SAMPLER_COSINE, 3, capi.ACCUMULATE_FLOAT, capi.FP16_OFF
        if a.mode == "closest":
            o.use_any_hit, o.falloff_distance, o.clamp_tmax_to_falloff = 0, 1.25, 1
        for k, v in zip(names, combo):
            setattr(o, k, v)
        with Baker(flat, o) as b:
            best = None
            for _ in range(a.reps + 1):
                b.build_bvh(); b.rasterize(0); b.trace(0); b.sync()
                s = b.last_stage_ms()
                if best is None or s["trace"] < best["trace"]:
                    best = s
            st = b.stats()
            crc = zlib.crc32(np.ascontiguousarray(b.results(0)).tobytes())
        print(json.dumps({"knobs": dict(zip(names, combo)), "bvh_build": round(best["bvh_build"] or 0, 3), "rasterize": round(best["rasterize"] or 0, 3), "trace": round(best["trace"], 3), "tile_cuts": round(best["tile_cuts"] or 0, 3),
                          "horizon": round(best["horizon"] or 0, 3), "walk": round(best["walk"] or 0, 3), "cast_share": round(st.rays_cast / max(1, st.ray_count), 4), "cells_crc32": crc}), flush=True)


if __name__ == "__main__":
    main()
