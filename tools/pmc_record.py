#!/usr/bin/env python3
"""tools/pmc_record.py PMC_DIR CALIB_DIR SLOPES.jsonl OUT.json — turns the rocprofv3 counter passes of one bench step
(tools/pmc_passes.sh -> PMC_DIR/summary.json), the known-byte kernels (tools/fetch_calib.hip run under `rocprofv3 --pmc
FETCH_SIZE`, CALIB_DIR) and the slope builds of the kernel (tools/build_variants.py valuN / loadN, tools/time_trace.py ->
SLOPES.jsonl) into the record bench.py reads (profiles/k_trace_counters.json):

  hbm_side_bytes_per_launch   FETCH_SIZE x 1 KiB x correction + WRITE_SIZE x 1 KiB, the correction being known bytes /
                              counted bytes of k_gather64 — a lane reading one random 64-byte record as 4 x 16 B, k_trace's
                              own access pattern (the guide's x2 is for wide streaming reads; measured here beside it)
  roof                        which unit binds k_trace, from measurements on THIS kernel:
      marginal_cost_per_vmem_instr   CU cycles one more vector-memory wave-instruction per inner visit costs the launch
                                     (slope of the load1 / load2 builds: (ms - base) x clock x CUs / extra wave-instructions);
                                     a 16-byte-per-lane load costs the CU's L1 path 4 + 0.65 x lanes cycles alone (tools/l1_probe.hip)
      marginal_cost_per_valu_instr   SIMD cycles one more v_xor_b32 per inner visit costs (valu8 / 16 / 32 builds); issued alone
                                     it costs 2.3 (tools/valu_probe.hip)
      ta_busy_frac, td_busy_frac     TA_BUSY_avr / TD_TD_BUSY_sum over the launch's shader cycles: the address and the data-return
                                     unit of the CU's vector-memory path
      bound                          "vmem" if a vector-memory instruction costs the launch more of its stand-alone price than a
                                     vector-ALU instruction does, else "valu"; frac = the busy fraction of that unit's counter
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

VALU_ALONE_CYCLES = 2.3          # v_xor_b32 / v_fma_f32, tools/valu_probe.hip (profiles/r03/roofs/valu_probe.jsonl)
CUS, SIMDS = 256, 1024


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


def slopes(path, clock_hz, inner_iterations):
    """marginal costs from the slope builds (any-hit lines of tools/time_trace.py's output)"""
    if not path or not os.path.exists(path):
        return None
    ms = {}
    for line in open(path):
        if not line.startswith("{"):
            continue
        r = json.loads(line)
        if r.get("mode") != "anyhit":
            continue
        name = os.path.basename(r["lib"]).replace("lib_", "").replace(".so", "")
        ms[name] = r["trace_ms"]
    if "base" not in ms:
        return None
    out = {"trace_ms": ms, "inner_iterations_per_launch": inner_iterations}
    valu = [(n, (ms[f"valu{n}"] - ms["base"]) * 1e-3 * clock_hz * SIMDS / (n * inner_iterations)) for n in (8, 16, 32) if f"valu{n}" in ms]
    vmem = [(n, (ms[f"load{n}"] - ms["base"]) * 1e-3 * clock_hz * CUS / (n * inner_iterations)) for n in (1, 2) if f"load{n}" in ms]
    if valu:
        out["marginal_cost_per_valu_instr"] = {"simd_cycles_by_extra_instr_per_visit": {str(n): round(c, 2) for n, c in valu},
                                               "mean": round(sum(c for _, c in valu) / len(valu), 2), "alone": VALU_ALONE_CYCLES}
    if vmem:
        out["marginal_cost_per_vmem_instr"] = {"cu_cycles_by_extra_instr_per_visit": {str(n): round(c, 2) for n, c in vmem},
                                               "mean": round(sum(c for _, c in vmem) / len(vmem), 2)}
    return out


def main():
    pmc_dir, calib_dir, slopes_path, out_path = sys.argv[1:5]
    import bench
    summary = json.load(open(os.path.join(pmc_dir, "summary.json")))
    c = summary["counters"]
    cal = calib(calib_dir)
    corr = cal.get("k_gather64", {}).get("known_over_counted")
    fetch_raw = c["FETCH_SIZE"] * 1024.0
    write = c["WRITE_SIZE"] * 1024.0
    xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    stats = summary.get("kernel_ms")                    # tools/pmc_summary.py: mean k_trace duration over the counter passes
    clock_hz = xcd_cycles / (stats * 1e-3) if stats else 2.38e9
    rays = 1073741824.0
    # inner-section executions of one launch: four 16-byte node loads each (the other vector-memory instructions: leaf triangles, frames)
    iterations = None
    lc = os.path.join(os.path.dirname(slopes_path) if slopes_path else "", "loop_counters.jsonl")
    if os.path.exists(lc):
        for line in open(lc):
            if line.startswith("{") and "loop_counters" in line:
                iterations = json.loads(line)["loop_counters"]["inner_sections"]
    if iterations is None:
        iterations = c.get("SQ_INSTS_VMEM_RD", 0.0) / 5.6      # (about 5.6 vector-memory instructions per inner iteration in all)
    sl = slopes(slopes_path, clock_hz, iterations)
    ta = c["TA_BUSY_avr"] / xcd_cycles if "TA_BUSY_avr" in c else None
    td = c["TD_TD_BUSY_sum"] / (xcd_cycles * CUS) if "TD_TD_BUSY_sum" in c else None
    roof = None
    if sl and "marginal_cost_per_vmem_instr" in sl and "marginal_cost_per_valu_instr" in sl:
        lanes = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"]
        vmem_alone = 4.0 + 0.65 * lanes * 0.65            # a dword load issued with the node's four: the L1 path's price for ~ the walking lanes
        share_vmem = sl["marginal_cost_per_vmem_instr"]["mean"] / 16.0
        share_valu = sl["marginal_cost_per_valu_instr"]["mean"] / VALU_ALONE_CYCLES
        bound = "vmem" if share_vmem >= share_valu else "valu"
        roof = {"bound": bound,
                "name": "the CU's vector-memory path (TA address unit / TD data return / L1): one lane's 16-byte request per node piece"
                        if bound == "vmem" else "vector-ALU issue",
                "frac": round(td if bound == "vmem" and td is not None else c["SQ_ACTIVE_INST_VALU"] / (xcd_cycles / 4.0 * SIMDS), 4),
                "frac_is": "TD_TD_BUSY_sum / (shader cycles x 256 CUs): the data-return unit of the vector-memory path" if bound == "vmem" else
                           "SQ_ACTIVE_INST_VALU / SIMD quad-cycles",
                "ta_busy_frac": round(ta, 4) if ta is not None else None, "td_busy_frac": round(td, 4) if td is not None else None,
                "marginal_cost_per_vmem_instr": sl["marginal_cost_per_vmem_instr"], "marginal_cost_per_valu_instr": sl["marginal_cost_per_valu_instr"],
                "reading": "one more 4-byte load per inner visit costs the launch %.1f CU cycles per wave-instruction — all of what the L1 path charges for it; "
                           "one more v_xor_b32 costs %.2f SIMD cycles of the %.1f it issues in: the vector-memory path binds, the vector ALU has slack"
                           % (sl["marginal_cost_per_vmem_instr"]["mean"], sl["marginal_cost_per_valu_instr"]["mean"], VALU_ALONE_CYCLES),
                "slope_builds_trace_ms": sl["trace_ms"]}
        _ = vmem_alone
    rec = {
        "kernel": summary["kernel"],
        "source_sha256_16": bench.kernel_source_hash(),
        "config": {"atlas": [2048, 2048], "spp": 256, "grid": 708, "n_gpus": 1, "bvh_width": 4, "bvh_quantised": 1, "bvh_builder": 2},
        "provenance": "rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cpp-host --no-wallclock` "
                      "(tools/pmc_passes.sh), summed over the k_trace dispatch; FETCH_SIZE corrected by the known-byte kernel k_gather64 "
                      "(tools/fetch_calib.hip); slopes from tools/build_variants.py + tools/time_trace.py; tools/pmc_record.py",
        "counters": c,
        "vgpr_count": summary.get("vgpr_count"), "lds_block_size": summary.get("lds_block_size"), "scratch_size": summary.get("scratch_size"),
        "fetch_size_calibration": cal,
        "fetch_correction_used": corr if corr else 1.0,
        "FETCH_bytes_raw": fetch_raw, "WRITE_bytes": write,
        "hbm_side_bytes_per_launch": fetch_raw * (corr if corr else 1.0) + write,
        "hbm_side_bytes_per_launch_raw": fetch_raw + write,
        "roof": roof,
        "clock_hz_in_pmc_passes": round(clock_hz),
        "valu_lane_occupancy": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]), 4),
        "valu_wave_instr_per_ray": round(c["SQ_INSTS_VALU"] / rays, 3),
        "salu_wave_instr_per_ray": round(c["SQ_INSTS_SALU"] / rays, 3) if "SQ_INSTS_SALU" in c else None,
        "vmem_wave_instr_per_ray": round(c["SQ_INSTS_VMEM_RD"] / rays, 3) if "SQ_INSTS_VMEM_RD" in c else None,
        "L2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in c else None,
        "L1_accesses_per_clock_per_cu": round(c["TCP_TOTAL_ACCESSES_sum"] / (xcd_cycles * CUS), 3) if "TCP_TOTAL_ACCESSES_sum" in c else None,
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
