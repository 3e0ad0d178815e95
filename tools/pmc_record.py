#!/usr/bin/env python3
"""tools/pmc_record.py PMC_DIR CALIB_DIR OUT.json — turns the rocprofv3 counter passes of one bench step
(tools/pmc_passes.sh -> PMC_DIR/summary.json) and of the known-byte kernels (tools/fetch_calib.hip run under
`rocprofv3 --pmc FETCH_SIZE`, CALIB_DIR) into the record bench.py reads (profiles/k_trace_counters.json):

  hbm_side_bytes_per_launch   FETCH_SIZE x 1 KiB x correction + WRITE_SIZE x 1 KiB, the correction being known bytes /
                              counted bytes of k_gather64 — a lane reading one random 64-byte record as 4 x 16 B, k_trace's
                              own access pattern (the guide's x2 is for wide streaming reads; measured here beside it)
  valu_issue_frac             SQ_ACTIVE_INST_VALU (quad-cycles a SIMD's VALU is occupied) / (GRBM_GUI_ACTIVE / 8 XCDs / 4
                              cycles per quad x 1024 SIMDs) — the roof that binds k_trace
  valu_lane_occupancy         SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)
and ties it to the kernel source (sha256 of trace.hip + aob_math.h + aob_internal.h) and the bench configuration, so that
bench.py reports it only for the kernel and workload it was measured on.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def calib(calib_dir):
    """known bytes / (FETCH_SIZE x 1024) per calibration kernel"""
    known = None
    for p in glob.glob(os.path.join(calib_dir, "*.log")):
        for line in open(p):
            if line.startswith("{"):
                known = json.loads(line)
    got = {}
    for path in glob.glob(os.path.join(calib_dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if row["Counter_Name"] == "FETCH_SIZE":
                name = row["Kernel_Name"]
                key = "k_stream" if "k_stream" in name else ("k_gather64" if "k_gather<4>" in name or "k_gatherILi4" in name else
                                                              ("k_gather48" if "k_gather<3>" in name or "k_gatherILi3" in name else None))
                if key:
                    got[key] = got.get(key, 0.0) + float(row["Counter_Value"])
    out = {}
    if known:
        for key, fetch in got.items():
            out[key] = {"known_bytes": known[key + "_bytes"], "FETCH_SIZE": fetch, "known_over_counted": known[key + "_bytes"] / (fetch * 1024.0)}
    return out


def main():
    pmc_dir, calib_dir, out_path = sys.argv[1:4]
    import bench
    summary = json.load(open(os.path.join(pmc_dir, "summary.json")))
    c = summary["counters"]
    cal = calib(calib_dir)
    corr = cal.get("k_gather64", {}).get("known_over_counted")
    fetch_raw = c["FETCH_SIZE"] * 1024.0
    write = c["WRITE_SIZE"] * 1024.0
    xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    rec = {
        "kernel": summary["kernel"],
        "source_sha256_16": bench.kernel_source_hash(),
        "config": {"atlas": [2048, 2048], "spp": 256, "grid": 708, "n_gpus": 1, "bvh_width": 4, "bvh_quantised": 1, "bvh_builder": 1},
        "provenance": "rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cpp-host --no-wallclock` "
                      "(tools/pmc_passes.sh), summed over the k_trace dispatch; FETCH_SIZE corrected by the known-byte kernel k_gather64 "
                      "(tools/fetch_calib.hip); tools/pmc_record.py",
        "counters": c,
        "vgpr_count": summary.get("vgpr_count"), "lds_block_size": summary.get("lds_block_size"), "scratch_size": summary.get("scratch_size"),
        "fetch_size_calibration": cal,
        "fetch_correction_used": corr if corr else 1.0,
        "FETCH_bytes_raw": fetch_raw, "WRITE_bytes": write,
        "hbm_side_bytes_per_launch": fetch_raw * (corr if corr else 1.0) + write,
        "hbm_side_bytes_per_launch_raw": fetch_raw + write,
        # (the counter sums the quad-cycles in which a WAVE has a VALU instruction in flight; instructions of different waves
        # overlap in the pipeline, and the two counters come from two passes: a saturated VALU reads slightly above 1)
        "valu_issue_frac": round(min(1.0, c["SQ_ACTIVE_INST_VALU"] / (xcd_cycles / 4.0 * 1024.0)), 4),
        "valu_issue_frac_uncapped": round(c["SQ_ACTIVE_INST_VALU"] / (xcd_cycles / 4.0 * 1024.0), 4),
        "valu_lane_occupancy": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]), 4),
        "valu_wave_instr_per_ray": round(c["SQ_INSTS_VALU"] / 1073741824.0, 3),
        "vmem_wave_instr_per_ray": round(c["SQ_INSTS_VMEM_RD"] / 1073741824.0, 3) if "SQ_INSTS_VMEM_RD" in c else None,
        "L2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in c else None,
        "shader_cycles_per_xcd_in_pmc_pass": xcd_cycles,
        "wave_time_share": ({"issuing": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                             "waiting_memory_or_barrier": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                             "ready_waiting_for_issue": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)}
                            if all(k in c for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")) else None),
    }
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
