#!/usr/bin/env python3
"""tools/scratch_report.py [MANGLED_KERNEL] — where the register spills of a k_trace instantiation sit.

Compiles aobaker_amd/csrc/trace.hip to gfx950 assembly with the library's flags and lists every scratch_load / scratch_store of the kernel
with the loop depth of its basic block (LLVM's own "Loop Header ... Depth=" / "in Loop: Header=... Depth=" annotations), the block's label,
and — for a spill inside the main loop — the first global / LDS instruction after it, which tells which path it feeds.  One JSON object.
Runs here (hipcc cross-compiles): python tools/scratch_report.py > profiles/r05/scratch_in_k_trace.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "_ZN3aob7k_traceILi1ELi1ELi1ELi11ELi2ELi0ELi1ELi1EEEvNS_11TraceParamsE"
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize"]
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-I", os.path.join(ROOT, "include"), "--save-temps", "-c", os.path.join(ROOT, "aobaker_amd", "csrc", "trace.hip"), "-o",
                        os.path.join(tmp, "t.o")], cwd=tmp, check=True, capture_output=True)
        text = open(os.path.join(tmp, "trace-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    start = text.index(want + ":")
    end = text.index(".end_amdhsa_kernel", start)
    body = text[start:end].splitlines()
    meta = text[text.index(".amdhsa_kernel " + want):end]
    depth, label = 0, "entry"
    spills, counts = [], {}
    for i, line in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            label, depth = m.group(1), 0
            for look in body[i:i + 4]:
                d = re.search(r"Depth=(\d+)", look)
                if d:
                    depth = int(d.group(1)); break
        elif re.search(r"; %bb\.\d+:", line):
            d = re.search(r"Depth=(\d+)", line)
            depth = int(d.group(1)) if d else (depth if "in Loop" in line else 0)
        if "scratch_" in line and not line.strip().startswith(";"):
            feeds = next((b.strip().split()[0] for b in body[i + 1:i + 14] if re.match(r"\s+(global_|buffer_|ds_|s_load)", b)), None)
            spills.append({"instruction": " ".join(line.split(";")[0].split()), "block": label, "loop_depth": depth, "next_memory_instruction": feeds})
            counts[depth] = counts.get(depth, 0) + 1
    rec = {"kernel": want, "private_segment_fixed_size": int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)),
           "vgprs": int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)),
           "scratch_instructions_by_loop_depth": {str(k): v for k, v in sorted(counts.items())},
           "loop_depths": "0: the kernel's prologue / epilogue; 1: once per work unit (tile x sample part: 4096 rays); 2: the walk's main loop (once per iteration); 3+: inside it",
           "spills": spills}
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
