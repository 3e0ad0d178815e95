#!/usr/bin/env python3
"""tools/pmc_record.py PMC_DIR CALIB_DIR SLOPES.jsonl OUT.json — turns the rocprofv3 counter passes of one bench step
(tools/pmc_passes.sh -> PMC_DIR/summary.json), the known-byte kernels (tools/fetch_calib.hip run under `rocprofv3 --pmc
FETCH_SIZE`, CALIB_DIR) and the slope builds of the kernel (tools/build_variants.py valuN / loadN, tools/time_trace.py ->
SLOPES.jsonl) into the record bench.py reads (profiles/k_trace_counters.json):

  fabric_side_bytes_per_launch  (requests leaving the L2s, Infinity-Cache hits included) FETCH_SIZE x 1 KiB x correction + WRITE_SIZE x 1 KiB, the correction being known bytes /
                              counted bytes of k_gather64 — a lane reading one random 64-byte record as 4 x 16 B, k_trace's
                              own access pattern (the guide's x2 is for wide streaming reads; measured here beside it)
  roof                        BOTH roofs of k_trace as throughput fractions, from measurements on THIS kernel, and the larger named:
      vmem.frac                      vector-memory wave-instructions (SQ_INSTS_VMEM_RD) x the marginal CU cycles of one of them (slope of
                                     the load1 / load2 builds: (ms - base) x clock x CUs / extra wave-instructions) / (shader cycles x 256 CUs)
      valu.frac                      sum over instruction classes of count x issue price / (shader cycles x 1024 SIMDs): 2.3 cycles for the
                                     FMA / MUL / ADD f32 and plain 32-bit integer classes, 4.2 for everything else (tools/valu_probe.hip),
                                     a mixed stream max(4.2 N_slow, 2.3 N_all); the class counts are the SQ_INSTS_VALU_* counters where
                                     the passes collected them, else SQ_INSTS_VALU at the kernel's static mix (tools/valu_mix.py)
      marginal costs                 what one more v_xor_b32 (valuN builds), one more v_perm_b32 (slowN builds) and one more dword load
                                     (loadN builds) per inner visit cost the launch
  valu_lane_occupancy         SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)
and ties it to the kernel source (sha256 of trace.hip + aob_math.h + aob_internal.h + bvh_build.hip) and to the configuration of the
PROFILED run: `config` and the ray count are copied from the JSON line that run printed (PMC_DIR/pass*.log: bench.py's
roofline.counter_key / rays_per_launch) — a record is refused when the passes carry no such line or disagree with each other —
so that bench.py reports it only for the kernel and workload it was measured on.
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
    slow = [(n, (ms[f"slow{n}"] - ms["base"]) * 1e-3 * clock_hz * SIMDS / (n * inner_iterations)) for n in (4, 8, 16) if f"slow{n}" in ms]
    if slow:
        out["marginal_cost_per_slow_valu_instr"] = {"simd_cycles_by_extra_instr_per_visit": {str(n): round(c, 2) for n, c in slow},
                                                    "mean": round(sum(c for _, c in slow) / len(slow), 2), "alone": 4.2, "instruction": "v_perm_b32"}
    vmem = [(n, (ms[f"load{n}"] - ms["base"]) * 1e-3 * clock_hz * CUS / (n * inner_iterations)) for n in (1, 2) if f"load{n}" in ms]
    if valu:
        out["marginal_cost_per_valu_instr"] = {"simd_cycles_by_extra_instr_per_visit": {str(n): round(c, 2) for n, c in valu},
                                               "mean": round(sum(c for _, c in valu) / len(valu), 2), "alone": VALU_ALONE_CYCLES}
    if vmem:
        out["marginal_cost_per_vmem_instr"] = {"cu_cycles_by_extra_instr_per_visit": {str(n): round(c, 2) for n, c in vmem},
                                               "mean": round(sum(c for _, c in vmem) / len(vmem), 2)}
    return out


def profiled_run(pmc_dir):
    """(configuration, rays per launch) of the run the counter passes profiled, from the JSON line each pass's program printed."""
    keys = []
    for p in sorted(glob.glob(os.path.join(pmc_dir, "pass*.log"))):
        line = None
        for text in open(p, errors="replace"):
            text = text.strip()
            if text.startswith("{") and '"roofline"' in text:
                try:
                    line = json.loads(text)
                except ValueError:
                    pass
        if line is None:
            continue
        r = line.get("roofline") or {}
        if r.get("counter_key") is None or not r.get("rays_per_launch"):
            continue
        keys.append((r["counter_key"], float(r["rays_per_launch"]), os.path.basename(p)))
    if not keys:
        raise SystemExit(f"{pmc_dir}: no pass log carries a bench line with roofline.counter_key / rays_per_launch — "
                         "the record would describe a configuration nobody measured; refused")
    for k, r, name in keys[1:]:
        if k != keys[0][0] or r != keys[0][1]:
            raise SystemExit(f"{pmc_dir}: {name} profiled {k} ({r:.0f} rays), {keys[0][2]} profiled {keys[0][0]} ({keys[0][1]:.0f} rays); refused")
    return keys[0][0], keys[0][1]


FAST_CLASSES = ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_INT32")


def valu_roof(c, xcd_cycles, static_mix):
    """The vector ALU's issue fraction: class counts x issue price / (cycles x SIMDs)."""
    n_all = c["SQ_INSTS_VALU"]
    if all(k in c for k in FAST_CLASSES):
        n_fast = sum(c[k] for k in FAST_CLASSES)
        how = "SQ_INSTS_VALU_{FMA,MUL,ADD}_F32 + INT32 counted as 2.3-cycle classes (an upper bound on them: INT32 holds shifts and compares too)"
    elif static_mix:
        n_fast = n_all * static_mix["fast_fraction"]
        how = f"SQ_INSTS_VALU at the kernel's static mix ({static_mix['fast_fraction']:.3f} of its vector instructions are 2.3-cycle ones: tools/valu_mix.py)"
    else:
        return None
    n_slow = n_all - n_fast
    cycles = max(4.2 * n_slow, 2.3 * n_all)
    return {"frac": round(cycles / (xcd_cycles * SIMDS), 4), "instructions": n_all, "of_which_2_3_cycle": round(n_fast),
            "price_model": "max(4.2 N_slow, 2.3 N_all) SIMD cycles (tools/valu_probe.hip)", "counts_from": how,
            "frac_if_every_instruction_cost_4_2": round(4.2 * n_all / (xcd_cycles * SIMDS), 4)}


def main():
    pmc_dir, calib_dir, slopes_path, out_path = sys.argv[1:5]
    static_mix = None
    if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
        static_mix = json.load(open(sys.argv[5]))
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
    key, rays = profiled_run(pmc_dir)
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
    if sl and "marginal_cost_per_vmem_instr" in sl:
        vm_cost = sl["marginal_cost_per_vmem_instr"]["mean"]
        vmem = {"frac": round(c["SQ_INSTS_VMEM_RD"] * vm_cost / (xcd_cycles * CUS), 4), "wave_instructions": c["SQ_INSTS_VMEM_RD"],
                "marginal_cu_cycles_per_instruction": sl["marginal_cost_per_vmem_instr"],
                "ta_busy_frac": round(ta, 4) if ta is not None else None, "td_busy_frac": round(td, 4) if td is not None else None,
                "note": "TA / TD busy are has-work counters, not throughput fractions; frac is instructions x their measured marginal cost"}
        valu = valu_roof(c, xcd_cycles, static_mix)
        if valu is not None:
            valu["marginal_cost_per_valu_instr"] = sl.get("marginal_cost_per_valu_instr")
            valu["marginal_cost_per_slow_valu_instr"] = sl.get("marginal_cost_per_slow_valu_instr")
        bound = "vmem" if valu is None or vmem["frac"] >= valu["frac"] else "valu"
        roof = {"bound": bound,
                "name": "the CU's vector-memory path (TA address unit / TD data return / L1): one lane's 16-byte request per node piece"
                        if bound == "vmem" else "vector-ALU issue (1024 SIMDs)",
                "frac": vmem["frac"] if bound == "vmem" else valu["frac"],
                "frac_is": "the larger of vmem.frac (vector-memory wave-instructions x their marginal CU cycles / (cycles x 256)) and valu.frac "
                           "(class counts x issue price / (cycles x 1024))",
                "vmem": vmem, "valu": valu, "slope_builds_trace_ms": sl["trace_ms"]}
    rec = {
        "kernel": summary["kernel"],
        "source_sha256_16": bench.kernel_source_hash(),
        "config": key, "rays_per_launch": rays,
        "provenance": "rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cpp-host --no-wallclock` "
                      "(tools/pmc_passes.sh), summed over the k_trace dispatch; FETCH_SIZE corrected by the known-byte kernel k_gather64 "
                      "(tools/fetch_calib.hip); slopes from tools/build_variants.py + tools/time_trace.py; tools/pmc_record.py",
        "counters": c,
        "vgpr_count": summary.get("vgpr_count"), "lds_block_size": summary.get("lds_block_size"), "scratch_size": summary.get("scratch_size"),
        "fetch_size_calibration": cal,
        "fetch_correction_used": corr if corr else 1.0,
        "FETCH_bytes_raw": fetch_raw, "WRITE_bytes": write,
        # requests leaving the L2s: Infinity-Cache (MALL) hits are counted in (MI355X_MICROARCH.md, HBM section) — fabric-side traffic, an
        # upper bound on what HBM itself moved (rounds 1-4 called it hbm_side_bytes_per_launch)
        "fabric_side_bytes_per_launch": fetch_raw * (corr if corr else 1.0) + write,
        "fabric_side_bytes_per_launch_raw": fetch_raw + write,
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
