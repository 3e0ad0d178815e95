#!/usr/bin/env python3
"""tools/valu_mix.py [SOURCE.hip] [MANGLED_KERNEL] — the static vector-ALU instruction mix of a kernel of the HIP library.

Compiles SOURCE (default: aobaker_amd/csrc/trace.hip) to gfx950 assembly with the library's flags and classifies the vector-ALU
instructions of MANGLED_KERNEL (default: the BASELINE configs[2] instantiation of k_trace) by issue price as tools/valu_probe.hip measured it: 2.3 cycles for v_fma_f32 / v_mul_f32 / v_add_f32 / v_sub_f32 /
v_fmac_f32 / v_xor_b32 / v_and_b32 / v_or_b32 / v_add_u32 / v_sub_u32 / v_mov_b32 with register operands, 4.2 for everything else
and for any instruction with a scalar-register or literal operand.  One JSON object; tools/pmc_record.py turns SQ_INSTS_VALU into
class counts with it where the counter passes have no per-class counters.  A STATIC mix: every instruction counts once, however
often it runs.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FAST = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_mov_b32"}


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "aobaker_amd", "csrc", "trace.hip")
    want = sys.argv[2] if len(sys.argv) > 2 else "_ZN3aob7k_traceILi1ELi1ELi1ELi11ELi2ELi0ELi1ELi1EEEvNS_11TraceParamsE"
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
             "-fno-slp-vectorize"]                                     # aobaker_amd/csrc/Makefile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o", out, src], check=True, capture_output=True)
        text = open(out).read()
    fast = slow = other = 0
    inside = False
    ops = {}
    for line in text.splitlines():
        if line.startswith(want + ":"):
            inside = True
            continue
        if inside and line.startswith(".Lfunc_end"):
            break
        if not inside:
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*(;.*)?$", line)
        if not m or m.group(1).startswith("."):
            continue
        op, args = m.group(1), m.group(2)
        if not op.startswith("v_"):
            other += 1
            continue
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        sources = args.split(",", 1)[1] if "," in args else ""
        scalar_operand = bool(re.search(r"(^|[ ,\[-])(s\d+|s\[\d+:\d+\]|vcc|exec|0x[0-9a-f]+)\b", sources))
        if base in FAST and not scalar_operand:
            fast += 1
        else:
            slow += 1
        ops[base] = ops.get(base, 0) + 1
    if not inside:
        raise SystemExit(f"no kernel {want!r} in the assembly of {src}")
    print(json.dumps({"kernel": want, "vector_alu_instructions": fast + slow, "fast_2_3_cycles": fast, "slow_4_2_cycles": slow,
                      "fast_fraction": round(fast / max(1, fast + slow), 4), "other_instructions": other,
                      "by_opcode": dict(sorted(ops.items(), key=lambda kv: -kv[1])[:24]), "note": "static: every instruction counted once"}))


if __name__ == "__main__":
    main()
