#!/usr/bin/env python3
"""Per-evaluation counter figures of the 20- / 61-state workloads (bench.py --config cfg3 / cfg4) from the pmc_sum.py files that
profiles/collect_generic.sh writes, tied to the profiled library by its sha256 -- bench.py's other_configs[].mfma reports counter-based
utilisation only for that very binary.

usage: generic_counters.py TAG OUTDIR > TAG_generic_counters.json      (reads OUTDIR/TAG_cfg{3,4}_pmc_*.json)
Every collection pass runs bench.py with --warmup 1 --steps 1: two evaluations per pass, so sums are halved.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 correction).
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EVALS_PER_PASS = 2.0


def total(path, counter):
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    vals = [k[counter]["sum"] for k in d.values() if counter in k]
    return sum(vals) / EVALS_PER_PASS if vals else None


def main():
    tag, out = sys.argv[1], sys.argv[2]
    with open(os.path.join(HERE, "physher_amd", "libphysher_amd.so"), "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    res = {"tag": tag, "library_sha256": sha, "evaluations_per_pass": EVALS_PER_PASS,
           "method": "rocprofv3 --pmc passes of bench.py --config cfgN --warmup 1 --steps 1 (profiles/collect_generic.sh), all k_lower_gen* / k_upper_gen* "
                     "launches summed and halved; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024"}
    for cfg in ("cfg3", "cfg4"):
        p = lambda name: os.path.join(out, f"{tag}_{cfg}_pmc_{name}.json")
        fetch, write = total(p("FETCH_SIZE"), "FETCH_SIZE"), total(p("WRITE_SIZE"), "WRITE_SIZE")
        res[cfg] = {
            "mfma_busy_cycles_per_eval": total(p("SQ_VALU_MFMA_BUSY_CYCLES"), "SQ_VALU_MFMA_BUSY_CYCLES"),
            "mfma_f64_instructions_per_eval": total(p("SQ_INSTS_VALU_MFMA_F64"), "SQ_INSTS_VALU_MFMA_F64"),
            "mfma_f64_mops_per_eval": total(p("SQ_INSTS_VALU_MFMA_F64"), "SQ_INSTS_VALU_MFMA_MOPS_F64"),
            "valu_instructions_per_eval": total(p("SQ_INSTS_VALU_MFMA_F64"), "SQ_INSTS_VALU"),
            "hbm_bytes_per_eval": None if fetch is None or write is None else (2.0 * fetch + write) * 1024.0,
        }
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
