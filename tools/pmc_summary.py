#!/usr/bin/env python3
"""Sum the rocprofv3 PMC counters of the k_trace launches under a tools/pmc_passes.sh output directory.

Prints one JSON object: counter name -> value summed over the k_trace dispatches of the pass that
collected it (one bench step = one launch per section), plus the launch count and kernel name.
"""
import csv
import glob
import json
import os
import sys


def main():
    root = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else "k_trace"
    out = {"counters": {}, "kernel": None, "launches": {}}
    # the instantiation measured: of the kernels whose name holds `kernel`, the one the passes spent the most time in — not the one-triangle
    # warm-up bake of aob_init (another instantiation, microseconds), nor a second mode the profiled program may have run
    spent = {}
    for path in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if kernel in row["Kernel_Name"]:
                    spent[row["Kernel_Name"]] = spent.get(row["Kernel_Name"], 0) + int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    if spent:
        kernel = max(spent, key=spent.get)
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        dispatches = set()
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if kernel not in row["Kernel_Name"] or (spent and row["Kernel_Name"] != kernel):
                    continue
                out["kernel"] = row["Kernel_Name"]
                name = row["Counter_Name"]
                out["counters"][name] = out["counters"].get(name, 0.0) + float(row["Counter_Value"])
                dispatches.add(row["Dispatch_Id"])
                out["lds_block_size"] = int(row["LDS_Block_Size"])
                out["vgpr_count"] = int(row["VGPR_Count"])
                out["scratch_size"] = int(row["Scratch_Size"])
        out["launches"][os.path.relpath(path, root)] = len(dispatches)
    # the kernel's duration in the counter passes (kernel-trace of the same runs): the clock the counters ran at = cycles / this
    # (per pass the SUM over that instantiation's launches — the sections of a bake — as the counters are summed; then the mean over the passes)
    ms = []
    for path in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        total = 0.0
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if kernel in row["Kernel_Name"] and (not spent or row["Kernel_Name"] == kernel):
                    total += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
        if total > 0.0:
            ms.append(total)
    if ms:
        out["kernel_ms"] = sum(ms) / len(ms)
        out["kernel_ms_per_pass"] = [round(x, 3) for x in ms]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
