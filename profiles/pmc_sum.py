#!/usr/bin/env python3
"""pmc_sum.py counter_collection.csv KERNEL_REGEX -- per kernel, every counter summed over all of the run's launches (and the
launch count): ratios between counters of one pass (busy / total cycles, MFMA / VALU instructions) need no per-evaluation split."""
import csv
import json
import re
import sys
from collections import defaultdict

rx = re.compile(sys.argv[2])
by = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if not rx.search(name):
        continue
    short = re.search(r"(k_[a-z0-9_]+)", name).group(1)
    e = by[short][r["Counter_Name"]]
    e[0] += float(r["Counter_Value"])
    e[1] += 1
print(json.dumps({k: {c: {"sum": v[0], "launches": v[1]} for c, v in ctrs.items()} for k, ctrs in by.items()}, indent=1))
