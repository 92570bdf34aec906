#!/usr/bin/env python3
"""Registers, spills, scratch and static LDS of the engine's kernels, read from the code object inside
physher_amd/libphysher_amd.so (the amdhsa.kernels notes the compiler wrote), keyed by the library's sha256.

usage: kernel_resources.py [regex] > out.json      (default regex: the tree-walk and MFMA kernels)
Dynamic LDS is a launch argument and not in the code object: see DESIGN.md (k_upper4_stream: 6400 B per wave x 4 waves).
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(HERE, "physher_amd", "libphysher_amd.so")


def main():
    rx = re.compile(sys.argv[1] if len(sys.argv) > 1 else r"k_(upper4_stream|upper4_walk|lower4_walk|lower_gen|upper_gen|lower_gen_walk|op_tables|slab_)")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "engine.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", LIB, os.path.join(d, "copy.so")], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == "group_segment_fixed_size":  # first field of a kernel's block (fields are sorted by name)
            cur = {"lds_static_bytes": int(val)}
        elif cur is not None and key == "name":
            cur["mangled"] = val
        elif cur is not None and key in ("private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "agpr_count"):
            cur[{"private_segment_fixed_size": "scratch_bytes"}.get(key, key)] = int(val)
        elif cur is not None and key == "wavefront_size":
            kernels[cur.pop("mangled")] = cur
            cur = None
    names = list(kernels)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    out = {}
    for mangled, name in zip(names, dem):
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"^void ", "", short).split("(")[0]
        if rx.search(short):
            out[short] = kernels[mangled]
    with open(LIB, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    json.dump({"library_sha256": sha, "kernels": out}, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
