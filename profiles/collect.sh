#!/bin/bash
# profiles/collect.sh TAG -- the three rocprofv3 passes behind bench.py's roofline object, run on the GPU box:
#   1. --kernel-trace --stats      : per-kernel durations (must agree with bench.py's HIP-event figures)
#   2. --pmc FETCH_SIZE            : HBM read traffic of the two hot kernels   (own pass, kernel trace only)
#   3. --pmc WRITE_SIZE            : HBM write traffic                           (own pass)
# Raw output goes to /tmp on the box; only the small filtered summaries are copied to gpurun_out/prof_TAG/
# (copy what is to be judged from there into profiles/).  Usage on the box, from the repo root:
#   bash profiles/collect.sh r01c
set -eo pipefail
TAG=${1:?tag}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PHYAMD_BENCH_BLOCK=1000000   # one generator block: fewer unrelated kernels in the trace
BENCH="python3 $ROOT/bench.py --no-cpu-baseline"
KREGEX='k_(lower4|upper4|lower_gen|upper_gen)'

rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $BENCH --steps 4 --warmup 1 > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/kt.err"
f=$(find /tmp/prof_kt -name 'kt_kernel_stats.csv' | head -1)
head -1 "$f" > "$OUT/${TAG}_kernel_stats.csv"
grep -E 'k_lower|k_upper|k_reduce_rows|k_transition|k_tip_tables|k_root' "$f" >> "$OUT/${TAG}_kernel_stats.csv" || true
f=$(find /tmp/prof_kt -name 'kt_kernel_trace.csv' | head -1)
head -1 "$f" > "$OUT/${TAG}_kernel_trace_phyamd.csv"
grep -E 'k_lower|k_upper|k_reduce_rows|k_transition|k_tip_tables|k_root' "$f" | tail -200 >> "$OUT/${TAG}_kernel_trace_phyamd.csv" || true
echo "kernel trace done" >&2

for ctr in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --pmc $ctr --kernel-trace --output-format csv --kernel-include-regex "$KREGEX" -d /tmp/prof_$ctr -o pmc -- $BENCH --steps 1 --warmup 1 > /dev/null 2> "$OUT/pmc_$ctr.err"
	f=$(find /tmp/prof_$ctr -name 'pmc_counter_collection.csv' | head -1)
	head -1 "$f" > "$OUT/${TAG}_pmc_$ctr.csv"
	grep -E "$KREGEX" "$f" | tail -8 >> "$OUT/${TAG}_pmc_$ctr.csv"
	echo "$ctr done" >&2
done
cd "$ROOT"
unset PHYAMD_BENCH_BLOCK
python3 bench.py --steps 5 --warmup 2 > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
ls -la "$OUT" >&2
