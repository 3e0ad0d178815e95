#!/usr/bin/env python3
"""tools/install_profile.py RUN_DIR [ROUND_DIR] — file the outputs of one `tools/profile_bench.sh RUN_DIR` run (merged back under gpurun_out/) where
the repo keeps them: profiles/ROUND_DIR/final/ (ROUND_DIR: default r05; record, raw sums, slopes, loop counters, kernel stats, shapes), profiles/k_trace_counters.json
(what bench.py reads) and profiles/strong_scaling_base.json (the one-GPU time of configs[3]'s atlas and of its strips, from the shapes of
the same run — tied to the kernel source hash of that run).  Run in the build container; then re-run `bench.py` on the GPU box so that the
committed bench line carries the record."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    run = os.path.join(ROOT, sys.argv[1])
    rnd = sys.argv[2] if len(sys.argv) > 2 else "r05"
    dst = os.path.join(ROOT, "profiles", rnd, "final")
    os.makedirs(os.path.join(dst, "roofs"), exist_ok=True)
    rows = [json.loads(l) for l in open(os.path.join(run, "shapes.jsonl"))]
    r1 = [r for r in rows if r["shape"] == "rank-1"][0]
    sh = json.load(open(os.path.join(run, "source_hash.json")))
    base = {"what": "single-GPU time of BASELINE configs[3]'s fixed atlas (build + rasterise + trace + resolve, scene resident): the base the N > 1 "
                    "strong-scaling values are read against; rank_strip_ms: what ONE GPU needed for a rank's strip of that atlas cut N ways in equal tile rows, "
                    "before any re-cut (first, middle and last rank) — the first real N-GPU run's rank_build_raster_trace is to be read against these",
            "atlas": 4096, "spp": 256, "n_gpus": 1, "bvh_builder": 2, "ms_per_step": r1["step_ms"], "value_rays_per_s": 4096 * 4096 * 256 / (r1["step_ms"] * 1e-3),
            "trace_kernel_ms": r1["trace_ms"], "rays": 4096 * 4096 * 256,
            "rank_strip_ms": {str(w): {str(r["rank"]): r["step_ms"] for r in rows if r["shape"] == f"rank-{w}"} for w in (2, 4, 8)},
            "command": f"python tools/run_shapes.py ranks  (profiles/{rnd}/final/shapes.jsonl)", "kernel_source_sha256_16": sh["source_sha256_16"],
            "measured": f"round {rnd[1:].lstrip('0')}, 1x MI355X, end-of-round kernel"}
    with open(os.path.join(ROOT, "profiles", "strong_scaling_base.json"), "w") as f:
        json.dump(base, f, indent=1)
        f.write("\n")
    for name in ("shapes.jsonl", "k_trace_counters.json", "source_hash.json", "bench.json", "stats_bench.json"):
        shutil.copy(os.path.join(run, name), os.path.join(dst, name))
    shutil.copy(os.path.join(run, "k_trace_counters.json"), os.path.join(ROOT, "profiles", "k_trace_counters.json"))
    shutil.copy(os.path.join(run, "pmc", "summary.json"), os.path.join(dst, "pmc_summary.json"))
    for path in glob.glob(os.path.join(run, "roofs", "*.jsonl")) + [os.path.join(run, "roofs", "valu_mix.json")]:
        shutil.copy(path, os.path.join(dst, "roofs", os.path.basename(path)))
    shutil.copy(glob.glob(os.path.join(run, "stats", "runc", "*_kernel_stats.csv"))[0], os.path.join(dst, "bench_kernel_stats.csv"))
    if os.path.exists(os.path.join(run, "knobs.jsonl")):
        shutil.copy(os.path.join(run, "knobs.jsonl"), os.path.join(dst, "knobs.jsonl"))
    if os.path.exists(os.path.join(run, "pytest.log")):
        with open(os.path.join(dst, "gpu_pytest_tail.log"), "w") as f:
            f.write("".join(open(os.path.join(run, "pytest.log")).readlines()[-5:]))
    print(sh["source_sha256_16"], base["ms_per_step"], base["rank_strip_ms"])


if __name__ == "__main__":
    main()
