#!/usr/bin/env python3
"""tools/build_variants.py NAME=FLAGS ... — A/B builds of the HIP library for kernel experiments.

Each NAME=FLAGS pair compiles trace.hip with the extra preprocessor FLAGS (e.g. `valu8=-DAOB_EXP_VALU=8`) and links it
with the product's other objects into tools/_variants/lib_NAME.so (git-ignored, shipped to the GPU box with the
snapshot).  tools/time_trace.py times them through AOBAKER_HIP_LIBRARY.  Runs here (hipcc cross-compiles).

  python tools/build_variants.py base= valu8=-DAOB_EXP_VALU=8 load1=-DAOB_EXP_LOAD=1
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aobaker_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_variants")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-Wall", "-Wno-unused-result", "-Wno-unused-value"]


def main():
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["make", "-s", "-j", "8", "-C", CSRC], check=True)
    # AOB_VARIANT_ALSO="bvh_build.hip ...": sources compiled with the variant's flags besides trace.hip (a flag that both sides of a
    # data structure must see, e.g. -DAOB_FAN=1)
    also = os.environ.get("AOB_VARIANT_ALSO", "").split()
    src_name = os.environ.get("AOB_VARIANT_REPLACES", os.environ.get("AOB_VARIANT_SOURCE", "trace.hip"))      # the product object the variant's source stands in for
    others = [os.path.join(CSRC, o) for o in ("api.o", "bvh_build.o", "raster.o", "trace.o", "horizon.o", "resolve.o") if o.replace(".o", ".hip") not in also + [src_name]]
    procs = []
    for arg in sys.argv[1:]:
        name, _, flags = arg.partition("=")
        src = os.environ.get("AOB_VARIANT_SOURCE", "trace.hip")
        objs = []
        for one in [src] + also:
            obj = os.path.join(OUT, f"{os.path.splitext(one)[0]}_{name}.o")
            cmd = [HIPCC] + FLAGS + flags.split() + ["-c", os.path.join(CSRC, one), "-o", obj]
            objs.append((obj, subprocess.Popen(cmd)))
        procs.append((name, objs))
    for name, objs in procs:
        for obj, p in objs:
            if p.wait() != 0:
                raise SystemExit(f"variant {name} failed to compile")
        lib = os.path.join(OUT, f"lib_{name}.so")
        subprocess.run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950"] + [o for o, _ in objs] + others + ["-o", lib], check=True)
        print(lib)


if __name__ == "__main__":
    main()
