#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes (separate runs, kernel trace only) into per-launch figures of the two tree-walk kernels:
HBM bytes from FETCH_SIZE / WRITE_SIZE and, optionally, vector-ALU wave-instructions from SQ_INSTS_VALU.

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counter values are KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a 16-B-per-lane coalesced stream, so it is doubled; WRITE_SIZE
is exact for 16-B-per-lane stores.  Only the launches of the LAST evaluation in the trace are used.

usage: pmc_traffic.py TAG FETCH.csv WRITE.csv taxa patterns categories launches_per_eval out.json [BUSY.csv]
launches_per_eval counts the pre-order kernel's launches (the chunked walk: 2); the post-order kernel's come from the
environment variable PMC_LOWER_LAUNCHES (default: the same number).
"""
import csv
import json
import os
import sys


def per_kernel(fn, key, launches, counter=None):
    rows = [r for r in csv.DictReader(open(fn)) if key in r["Kernel_Name"] and (counter is None or r["Counter_Name"] == counter)]
    rows = rows[-launches:]
    return sum(float(r["Counter_Value"]) for r in rows), len(rows)


def library_hash():
    """sha256 of the engine library the profiled process loaded: bench.py reports roofline.frac only for the same binary"""
    import hashlib
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(here, "physher_amd", "libphysher_amd.so"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def kernel_resources(fn, key):
    """registers, spills and scratch of the kernel that ran (its name from the profiler's rows), read from the code object of the
    profiled library by profiles/kernel_resources.py; workgroup size as the profiler saw the launch"""
    import re
    import subprocess
    for r in csv.DictReader(open(fn)):
        if key in r["Kernel_Name"]:
            m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"]
            here = os.path.dirname(os.path.abspath(__file__))
            try:
                table = json.loads(subprocess.run([sys.executable, os.path.join(here, "kernel_resources.py"), "k_"], check=True, capture_output=True, text=True).stdout)["kernels"]
            except Exception as exc:  # (llvm tools missing: keep the name)
                return {"kernel": name, "error": str(exc)}
            res = dict(table.get(name, {}))
            res.update({"kernel": name, "workgroup": int(r["Workgroup_Size"])})
            return res
    return None


def main():
    tag, fetch, write = sys.argv[1], sys.argv[2], sys.argv[3]
    T, P, C, L, out = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), sys.argv[8]
    busy = sys.argv[9] if len(sys.argv) > 9 else None
    LL = int(os.environ.get("PMC_LOWER_LAUNCHES", L))
    res = {"tag": tag, "taxa": T, "patterns": P, "categories": C, "states": 4, "launches_per_eval": L, "lower_launches_per_eval": LL, "source": [fetch, write] + ([busy] if busy else []),
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction); "
                     "VALU = SQ_INSTS_VALU (wave-instructions) of the same launches in a third pass"}
    res["library_sha256"] = library_hash()
    res["kernels"] = {"upper": kernel_resources(fetch, "k_upper4"), "lower": kernel_resources(fetch, "k_lower4")}
    for name, key, L in (("upper", "k_upper4", L), ("lower", "k_lower4", LL)):
        f, n1 = per_kernel(fetch, key, L)
        w, n2 = per_kernel(write, key, L)
        assert n1 == n2 == L, (n1, n2, L)
        f *= 1024.0
        w *= 1024.0
        res[f"{name}_fetch_bytes_raw_per_eval"] = f
        res[f"{name}_write_bytes_per_eval"] = w
        res[f"{name}_bytes_per_eval"] = 2 * f + w
        res[f"{name}_bytes_per_launch"] = (2 * f + w) / L
        if busy:
            for ctr in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS"):
                v, n3 = per_kernel(busy, key, L, ctr)
                if n3 == L:
                    res[f"{name}_{ctr[9:].lower()}_insts_per_launch"] = v / L
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
