#!/usr/bin/env python3
"""tools/time_trace.py [--modes anyhit,closest,cfg2,cfg5s] [--reps 4] LIB ... — k_trace time of A/B builds of the library.

One child process per LIB (a library is loaded once per process, through AOBAKER_HIP_LIBRARY); each builds the BVH,
rasterises, traces `reps + 1` times and prints one JSON line per mode: best trace time (HIP events on the library's own
stream, aob_bake_last_trace_ms) and the CRC-32 of the result cells — equal CRCs across builds mean equal results.
Product code only (no oracle).  LIB `-` is the in-tree library.

  anyhit   BASELINE configs[2]: 999 698-triangle terrain, 2048^2, 256 spp, cosine / any hit / float   k_trace<1,1,1,11,2,0>
  closest  the same terrain and rays with closest-hit falloff scoring (falloff 1.25, tmax clamped)      k_trace<1,0,1,10,2,0>
  refmode  the reference's own constants on the same terrain: 100 spp, table sampler, int32 sums, any hit           k_trace<0,1,0,10,2,0>
  cfg2     configs[1]: 69 938-triangle sphere, 1024^2, 64 spp, any hit
  cfg5     configs[4]'s instanced scene (10 M triangles) at its own 8192^2 atlas, 16 of its 1024 spp, closest-hit falloff scoring
  cfg5full configs[4] itself: the instanced scene at 8192^2 and all 1024 spp (58 G rays, ~10 s per trace: use --reps 1)
  cfg5s    configs[4]'s instanced scene (10 M triangles) at 2048^2 x 64 spp, closest-hit falloff scoring

  python tools/build_variants.py base= valu8=-DAOB_EXP_VALU=8 && python tools/time_trace.py tools/_variants/lib_*.so
"""
import argparse
import json
import os
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(modes, reps, knobs):
    import numpy as np
    import torch  # noqa: F401  (one HIP runtime)
    from aobaker_amd import BakeEngine, Baker, capi, scenes
    from aobaker_amd.scene import FlatScene
    BakeEngine.Init(0)
    lib = os.environ.get("AOBAKER_HIP_LIBRARY", "-")

    def cosine(spp, seed, **kw):
        o = capi.default_options()
        o.sample_count, o.sampler, o.accumulate, o.normal_fp16, o.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, seed
        for k, v in list(kw.items()) + list(knobs.items()):
            setattr(o, k, v)
        return o

    terrain = None
    for mode in modes:
        if mode in ("anyhit", "closest"):
            terrain = terrain or FlatScene(scenes.terrain_scene(grid=708, atlas=2048, seed=3, bpc=4))
            flat, spp = terrain, 256
            o = cosine(256, 3) if mode == "anyhit" else cosine(256, 3, use_any_hit=0, falloff_distance=1.25, clamp_tmax_to_falloff=1)
        elif mode == "refmode":
            # the reference's own constants (BakeEngine.cpp:13-18): 100 spp, MSVC-rand table, int32 quantised sums, any hit, R8
            terrain1 = FlatScene(scenes.terrain_scene(grid=708, atlas=2048, seed=3, bpc=1))
            flat, spp = terrain1, 100
            o = capi.default_options()
            for k, v in knobs.items():
                setattr(o, k, v)
        elif mode == "cfg2":
            flat, spp = FlatScene(scenes.sphere_scene(n=187, atlas=1024, seed=2, bpc=4)), 64
            o = cosine(64, 2)
        elif mode == "cfg5":
            flat, spp = FlatScene(scenes.instanced_scene(instances=100, rock_n=224, atlas=8192, seed=5, bpc=1)), 16
            o = cosine(16, 5, use_any_hit=0, falloff_distance=1.25, clamp_tmax_to_falloff=1)
        elif mode == "cfg5full":
            flat, spp = FlatScene(scenes.instanced_scene(instances=100, rock_n=224, atlas=8192, seed=5, bpc=1)), 1024
            o = cosine(1024, 5, use_any_hit=0, falloff_distance=1.25, clamp_tmax_to_falloff=1)
        elif mode == "cfg5s":
            flat, spp = FlatScene(scenes.instanced_scene(instances=100, rock_n=224, atlas=2048, seed=5, bpc=1)), 64
            o = cosine(64, 5, use_any_hit=0, falloff_distance=1.25, clamp_tmax_to_falloff=1)
        else:
            raise SystemExit(f"unknown mode {mode}")
        with Baker(flat, o) as b:
            b.build_bvh()
            b.rasterize(0)
            ms = []
            for _ in range(reps + 1):
                b.clear_results(0)
                b.trace(0)
                b.sync()
                ms.append(b.last_trace_ms())
            n = b.surfel_count(0)
            crc = zlib.crc32(np.ascontiguousarray(b.results(0)).tobytes())
            import time as _time
            build_ms = []
            for _ in range(3):                                   # the build's wall time, three times (the stream drained before and after)
                b.sync(); t0 = _time.perf_counter(); b.build_bvh(); b.sync(); build_ms.append((_time.perf_counter() - t0) * 1e3)
            st = b.stats()
            stats = None
            if hasattr(b.lib, "aob_debug_trace_stats"):          # a diagnostic build (-DAOB_TRACE_STATS=1): the loop counters of ONE trace
                import ctypes
                buf = (ctypes.c_ulonglong * 16)()
                b.lib.aob_debug_trace_stats(buf)
                b.clear_results(0); b.trace(0); b.sync()
                b.lib.aob_debug_trace_stats(buf)
                names = ("loop_iterations", "inner_sections", "inner_lanes", "leaf_sections", "leaf_lanes", "refill_sections", "pops", "generation_batches",
                         "cut_entries_tested", "cut_entries_entered", "drain_packed", "deep_stack_visits")
                stats = {k: int(buf[i]) for i, k in enumerate(names)}
                stats["hbm_stack_pushes"], stats["hbm_stack_pops"] = int(buf[12]), int(buf[13])      # (counted only by experiment builds; 0 otherwise)
                # (inner sections run after the unit's last ray was generated: sections << 40 | lanes)
                stats["drain_inner_sections"], stats["drain_inner_lanes"] = stats["drain_packed"] >> 40, stats["drain_packed"] & ((1 << 40) - 1)
                del stats["drain_packed"]
                rays = n * spp
                stats["per_ray"] = {"loop_iterations": round(stats["loop_iterations"] / rays, 4), "inner_visits": round(stats["inner_lanes"] / rays, 3),
                                    "leaf_visits": round(stats["leaf_lanes"] / rays, 3),
                                    "lanes_per_inner_section": round(stats["inner_lanes"] / max(1, stats["inner_sections"]), 2),
                                    "lanes_per_leaf_section": round(stats["leaf_lanes"] / max(1, stats["leaf_sections"]), 2),
                                    "pops_per_refill": round(stats["pops"] / max(1, stats["refill_sections"]), 2),
                                    "inner_sections_with_a_deep_stack": round(stats["deep_stack_visits"] / max(1, stats["inner_sections"]), 4),
                                    "inner_sections_in_the_drain": round(stats["drain_inner_sections"] / max(1, stats["inner_sections"]), 4),
                                    "lanes_per_inner_section_in_the_drain": round(stats["drain_inner_lanes"] / max(1, stats["drain_inner_sections"]), 2)}
            hz = None
            if hasattr(b.lib, "aob_bake_horizon_counters") and st.trace_horizon == 1:
                import ctypes
                hb = (ctypes.c_ulonglong * 16)()
                b.lib.aob_bake_horizon_counters(b.handle, hb, 2)
                b.clear_results(0); b.trace(0); b.sync()
                b.lib.aob_bake_horizon_counters(b.handle, hb, 1)
                tl = max(1, int(hb[5]))
                hz = {"tiles": int(hb[5]), "per_tile": {"rounds": round(hb[0] / tl, 1), "box_tests": round(hb[1] / tl, 1), "near_triangles": round(hb[2] / tl, 1),
                                                        "middle_leaves": round(hb[3] / tl, 1), "given_up": round(hb[4] / tl, 4)},
                      "clock_ticks_per_tile": {"walk": round(hb[6] / tl), "near": round(hb[7] / tl), "arcs_and_out": round(hb[8] / tl)},
                      "rays_cast_share": round(st.rays_cast / max(1, st.ray_count), 4)}
        best = min(ms[1:])
        print(json.dumps({"lib": os.path.relpath(lib, ROOT) if lib != "-" else "in-tree", "mode": mode, "knobs": knobs, "trace_ms": round(best, 3),
                          "all_ms": [round(x, 3) for x in ms], "grays_per_s": round(n * spp / best / 1e6, 3), "results_crc32": crc,
                          "bvh_build_wall_ms": round(min(build_ms), 3), "bvh_depth": st.bvh_depth, **({"horizon": hz} if hz else {}), **({"loop_counters": stats} if stats else {})}), flush=True)
    BakeEngine.Dispose()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="anyhit")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--knob", action="append", default=[], help="AobOptions field=value (e.g. trace_refill_idle=24)")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("libs", nargs="*", default=["-"])
    a = ap.parse_args()
    knobs = {k: int(v) for k, v in (kv.split("=") for kv in a.knob)}
    if a.child:
        child(a.modes.split(","), a.reps, knobs)
        return
    for lib in a.libs or ["-"]:
        env = dict(os.environ)
        if lib != "-":
            env["AOBAKER_HIP_LIBRARY"] = os.path.abspath(lib)
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--modes", a.modes, "--reps", str(a.reps)] + [f"--knob={k}" for k in a.knob]
        r = subprocess.run(cmd, env=env, timeout=900)
        if r.returncode != 0:
            print(json.dumps({"lib": lib, "error": f"child exited {r.returncode}"}), flush=True)


if __name__ == "__main__":
    main()
