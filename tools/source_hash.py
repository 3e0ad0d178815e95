#!/usr/bin/env python3
"""tools/source_hash.py [--manifest DIR] — which kernel source a measurement belongs to.

Without arguments: one JSON object with the SHA-256 (first 16 hex digits) of every device source of the library
(aobaker_amd/csrc/*.hip, *.h), of them all together (`kernel_source_sha`) and of the files k_trace's time depends on as
bench.py hashes them (`source_sha256_16`, what bench.py's line and profiles/k_trace_counters.json carry): tools/profile_bench.sh
stores it next to what it measures.

--manifest DIR (run in the build container, needs git): writes DIR/MANIFEST.json — for every file under DIR the commit that
added or last changed it and the `kernel_source_sha` of the sources AT that commit, i.e. the kernel the numbers in the file were
measured on (measurement files are committed together with, or right after, the source they measured)."""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = "aobaker_amd/csrc"


def digest(blobs):
    per = {name: hashlib.sha256(data).hexdigest()[:16] for name, data in blobs}
    h = hashlib.sha256()
    for name in sorted(per):
        h.update(name.encode() + b"\0" + per[name].encode() + b"\n")
    k = hashlib.sha256()
    for rel in ("aobaker_amd/csrc/trace.hip", "aobaker_amd/csrc/aob_math.h", "aobaker_amd/csrc/aob_internal.h", "aobaker_amd/csrc/bvh_build.hip",
                "aobaker_amd/csrc/horizon_kernel.h", "aobaker_amd/csrc/horizon.hip"):      # bench.py KERNEL_SOURCES
        k.update(dict(blobs).get(rel, b""))
    return {"kernel_source_sha": h.hexdigest()[:16], "source_sha256_16": k.hexdigest()[:16], "files": per}


def working_tree():
    names = sorted(glob.glob(os.path.join(ROOT, CSRC, "*.hip")) + glob.glob(os.path.join(ROOT, CSRC, "*.h")))
    return digest([(os.path.relpath(n, ROOT), open(n, "rb").read()) for n in names])


def at_commit(commit):
    names = subprocess.run(["git", "ls-tree", "--name-only", commit, CSRC + "/"], cwd=ROOT, capture_output=True, text=True, check=True).stdout.split()
    names = [n for n in names if n.endswith((".hip", ".h"))]
    return digest([(n, subprocess.run(["git", "show", f"{commit}:{n}"], cwd=ROOT, capture_output=True, check=True).stdout) for n in names])


def manifest(directory):
    out = {}
    cache = {}
    for path in sorted(glob.glob(os.path.join(ROOT, directory, "**", "*"), recursive=True)):
        rel = os.path.relpath(path, ROOT)
        if os.path.isdir(path) or rel.endswith("MANIFEST.json"):
            continue
        commit = subprocess.run(["git", "log", "-1", "--format=%H", "--", rel], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        if not commit:
            out[os.path.relpath(path, os.path.join(ROOT, directory))] = {"commit": None, "note": "not committed yet: the working tree's sources", **{"kernel_source_sha": working_tree()["kernel_source_sha"]}}
            continue
        if commit not in cache:
            h = at_commit(commit)
            cache[commit] = {"kernel_source_sha": h["kernel_source_sha"], "source_sha256_16": h["source_sha256_16"]}
        out[os.path.relpath(path, os.path.join(ROOT, directory))] = {"commit": commit[:12], **cache[commit]}
    with open(os.path.join(ROOT, directory, "MANIFEST.json"), "w") as f:
        json.dump({"what": "file -> the commit that last changed it, the SHA of aobaker_amd/csrc/*.{hip,h} at that commit (kernel_source_sha) and of the files k_trace's time depends on as bench.py hashes them (source_sha256_16; before round 3's end that list had no bvh_build.hip) — tools/source_hash.py",
                   "files": out}, f, indent=1)
        f.write("\n")
    print(f"{len(out)} files -> {directory}/MANIFEST.json")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--manifest":
        manifest(sys.argv[2])
    else:
        print(json.dumps(working_tree()))
