#!/bin/bash
# profiles/level_trace.sh cfgN -- per-launch durations (us) of one evaluation of a 20- / 61-state workload, in launch order, with the
# gaps between launches: where a level-scheduled pass loses its time (a diagnostic; output on stdout)
set -o pipefail
CFG=${1:?cfg3|cfg4}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_$CFG -o kt -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --no-distinct-check --no-drop-in --config $CFG --steps 2 --warmup 1 > /dev/null 2> /tmp/lt_$CFG.err
f=$(find /tmp/lt_$CFG -name 'kt_kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last evaluation: from the last k_transition_matrices on
idx = max(i for i, r in enumerate(rows) if "k_transition_matrices" in r["Kernel_Name"])
ev = rows[idx:]
t0 = int(ev[0]["Start_Timestamp"])
prev_end = None
tot = 0
for r in ev:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:26]
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  {name:26s} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}x{r.get('Grid_Size_Y', '')} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}")
    prev_end = e
    tot += e - s
print(f"sum of kernel durations {tot / 1e3:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")
PY
