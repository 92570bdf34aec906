#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into per-launch HBM bytes.

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counter values are KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a 16-B-per-lane coalesced stream, so it is doubled; WRITE_SIZE
is exact for 16-B-per-lane stores.  Only the launches of the LAST evaluation in the trace are used.

usage: pmc_traffic.py FETCH.csv WRITE.csv taxa patterns categories launches_per_eval out.json
"""
import csv
import json
import sys


def per_kernel(fn, key, launches):
    rows = [r for r in csv.DictReader(open(fn)) if key in r["Kernel_Name"]]
    rows = rows[-launches:]
    return sum(float(r["Counter_Value"]) for r in rows) * 1024.0, len(rows)


def main():
    fetch, write, T, P, C, L, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    res = {"taxa": T, "patterns": P, "categories": C, "launches_per_eval": L, "source": [fetch, write],
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction)"}
    for name, key in (("upper", "k_upper4"), ("lower", "k_lower4")):
        f, n1 = per_kernel(fetch, key, L)
        w, n2 = per_kernel(write, key, L)
        assert n1 == n2 == L, (n1, n2, L)
        res[f"{name}_fetch_bytes_raw_per_eval"] = f
        res[f"{name}_write_bytes_per_eval"] = w
        res[f"{name}_bytes_per_eval"] = 2 * f + w
        res[f"{name}_bytes_per_launch"] = (2 * f + w) / L
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
