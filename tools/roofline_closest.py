#!/usr/bin/env python3
"""tools/roofline_closest.py DIR OUT.json — the roofline object of the CLOSEST-HIT instantiation of k_trace on BASELINE configs[4]'s scene
(VERDICT r4 item 2), assembled from what tools/profile_cfg5.sh left in DIR on the GPU box:

  time_builder1.jsonl      tools/time_trace.py --modes cfg5 (trace ms of one launch: 10 M triangles, 8192^2 atlas, 16 of the 1024 spp)
  loop_counters.jsonl      the same launch on the -DAOB_TRACE_STATS=1 build: node visits, leaf visits, entries tested — counted by the kernel itself
  pmc/summary.json         rocprofv3 --pmc passes over that program, summed over its k_trace launches (tools/pmc_summary.py)
  stats/.../kernel_stats   the rocprofv3 --kernel-trace --stats summary (copied beside this record)

The same bounded fractions as bench.py's roofline: bytes the kernel's lanes request from the vector L1 against 64 B / clk / CU (the binding
unit of the any-hit instantiation; for this one the record says which of its numbers is largest), L1 -> L2 requests against L2 peak,
fabric-side bytes against HBM peak, and the instruction rates per ray.  Runs anywhere (no GPU): python tools/roofline_closest.py gpurun_out/cfg5 profiles/r05/cfg5/roofline.json
"""
import json
import os
import sys

L1_PEAK_GBS = 64.0 * 256 * 2.4
L2_PEAK_GBS = 34500.0
HBM_PEAK_GBS = 8000.0


def last_json(path):
    with open(path) as f:
        lines = [l for l in f.read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def main():
    d, out = sys.argv[1], sys.argv[2]
    t = last_json(os.path.join(d, "time_builder1.jsonl"))
    lc = last_json(os.path.join(d, "loop_counters.jsonl"))["loop_counters"]
    pmc = json.load(open(os.path.join(d, "pmc", "summary.json")))
    c = pmc["counters"]
    rays = lc["pops"]
    per = lc["per_ray"]
    seconds = t["trace_ms"] * 1e-3
    # the launches the counters were summed over: the profiled program traces twice (tools/time_trace.py --reps 1); the summed kernel time says so
    # (the dispatch count of a pass also holds what other instantiations the process ran)
    launches = max(1, round(pmc["kernel_ms"] / t["trace_ms"]))
    # what a ray asks the vector L1 for: 64 B per node visit, 48 B per triangle of a leaf visit (a leaf slot is a pair of triangles or one: between 48 and
    # 96 B — the pair share is not counted by the kernel, both bounds are given), 16 B of its surfel's record at the pop, 48 B of the frame at its generation
    lo = 64.0 * per["inner_visits"] + 48.0 * per["leaf_visits"] + 64.0
    hi = 64.0 * per["inner_visits"] + 96.0 * per["leaf_visits"] + 64.0
    req_lo, req_hi = rays * lo / seconds / 1e9, rays * hi / seconds / 1e9
    fetch_per_launch = c["FETCH_SIZE"] * 1024.0 / launches
    l2_per_launch = c["TCP_TCC_READ_REQ_sum"] * 64.0 / launches
    pmc_seconds = pmc["kernel_ms"] * 1e-3 / launches
    rec = {
        "kernel": pmc["kernel"], "workload": "BASELINE configs[4]'s scene: 10 035 202 triangles, 8192^2 atlas, COSINE sampler, closest-hit falloff scoring, tmax = falloff = 1.25; "
                                             "one launch of 16 of its 1024 spp (tools/time_trace.py cfg5)",
        "rays_per_launch": rays, "trace_ms": t["trace_ms"], "grays_per_s": t["grays_per_s"], "results_crc32": t["results_crc32"],
        "per_ray": {"node_visits": per["inner_visits"], "leaf_visits": per["leaf_visits"], "loop_iterations": per["loop_iterations"],
                    "cut_entries_tested": round(lc["cut_entries_tested"] * 64.0 / max(1, rays), 2),
                    "lanes_per_inner_section": per["lanes_per_inner_section"], "lanes_per_leaf_section": per["lanes_per_leaf_section"],
                    "valu_wave_instr": round(c["SQ_INSTS_VALU"] / launches / rays, 2), "vmem_wave_instr": round(c["SQ_INSTS_VMEM_RD"] / launches / rays, 3),
                    "salu_wave_instr": round(c["SQ_INSTS_SALU"] / launches / rays, 2)},
        "bound": "vmem", "unit": "GB/s", "peak": round(L1_PEAK_GBS, 1),
        "achieved": round(req_hi, 1), "frac": round(req_hi / L1_PEAK_GBS, 4),
        "frac_is": "bytes the lanes request from the vector L1 (64 B per node visit, 96 B per leaf visit — every leaf a pair: the upper bound —, 64 B per ray of surfel "
                   "record and frame) / kernel time / (64 B x 256 CUs x 2.4 GHz)",
        "frac_if_no_leaf_were_a_pair": round(req_lo / L1_PEAK_GBS, 4),
        "l2": {"bytes_per_launch": l2_per_launch, "frac": round(l2_per_launch / pmc_seconds / 1e9 / L2_PEAK_GBS, 4), "L2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4),
               "note": "TCP_TCC_READ_REQ_sum x 64 B against 34.5 TB/s"},
        "fabric": {"bytes_per_launch": fetch_per_launch, "frac": round(fetch_per_launch / pmc_seconds / 1e9 / HBM_PEAK_GBS, 4),
                   "note": "FETCH_SIZE x 1 KiB (uncorrected; Infinity-Cache hits included) against 8 TB/s"},
        "valu": {"wave_instructions_per_launch": c["SQ_INSTS_VALU"] / launches,
                 "frac_at_3_cycles_per_instruction": round(c["SQ_INSTS_VALU"] / launches * 3.0 / (pmc_seconds * 2.2e9 * 1024), 4),
                 "note": "vector-ALU wave-instructions x ~3 SIMD cycles (the any-hit kernel's measured mix: max(4.2 N_slow, 2.3 N_all)) / (kernel time x 2.2 GHz x 1024 SIMDs): "
                         "a rough issue fraction — this instantiation has no slope builds of its own"},
        "source": {"time": t, "pmc_kernel_ms_per_pass": pmc.get("kernel_ms_per_pass"), "pmc_launches_per_pass": launches},
    }
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print(json.dumps({k: rec[k] for k in ("kernel", "trace_ms", "grays_per_s", "frac", "per_ray")}))


if __name__ == "__main__":
    main()
