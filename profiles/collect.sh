#!/bin/bash
# profiles/collect.sh TAG -- the rocprofv3 passes behind bench.py's roofline object, run on the GPU box:
#   1. --kernel-trace --stats      : per-kernel durations (must agree with bench.py's HIP-event figures)
#   2. --pmc FETCH_SIZE            : HBM read traffic of the two hot kernels   (own pass, kernel trace only)
#   3. --pmc WRITE_SIZE            : HBM write traffic                           (own pass)
#   4. --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS : instruction mix (own pass)
#   5. --pmc VALUBusy SALUBusy MemUnitBusy MemUnitStalled / SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
# Raw output goes to /tmp on the box; the small filtered summaries land in gpurun_out/prof_TAG/, together with TAG_traffic.json
# (per-launch HBM bytes, VALU wave-instructions, the kernels' registers / LDS and the sha256 of the profiled libphysher_amd.so:
# copy it to profiles/traffic_latest.json for bench.py's roofline object, which reports frac only for that very binary) --
# nothing under profiles/ is written here; copy what is to be judged from gpurun_out/prof_TAG/ into profiles/.  Usage on the box, from the repo root:   bash profiles/collect.sh r02a
set -eo pipefail
TAG=${1:?tag}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PHYAMD_BENCH_BLOCK=1000000   # one generator block: fewer unrelated kernels in the trace
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --no-distinct-check"
KREGEX='k_(lower4|upper4|lower_gen|upper_gen)'

rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $BENCH --steps 4 --warmup 1 > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/kt.err"
f=$(find /tmp/prof_kt -name 'kt_kernel_stats.csv' | head -1)
head -1 "$f" > "$OUT/${TAG}_kernel_stats.csv"
grep -E 'k_lower|k_upper|k_reduce_rows|k_transition|k_tip_tables|k_root|k_op_tables|k_slab|k_build_mask' "$f" >> "$OUT/${TAG}_kernel_stats.csv" || true
f=$(find /tmp/prof_kt -name 'kt_kernel_trace.csv' | head -1)
head -1 "$f" > "$OUT/${TAG}_kernel_trace_phyamd.csv"
grep -E 'k_lower|k_upper|k_reduce_rows|k_transition|k_tip_tables|k_root|k_op_tables|k_slab|k_build_mask' "$f" | tail -40 >> "$OUT/${TAG}_kernel_trace_phyamd.csv" || true
echo "kernel trace done" >&2

pass=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "VALUBusy SALUBusy MemUnitBusy MemUnitStalled" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
            "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM"; do
	pass=$((pass + 1))
	name=$(echo $ctrs | cut -d' ' -f1)
	rocprofv3 --pmc $ctrs --kernel-trace --output-format csv --kernel-include-regex "$KREGEX" -d /tmp/prof_p$pass -o pmc -- $BENCH --steps 1 --warmup 1 > /dev/null 2> "$OUT/pmc_$name.err"
	f=$(find /tmp/prof_p$pass -name 'pmc_counter_collection.csv' | head -1)
	head -1 "$f" > "$OUT/${TAG}_pmc_$name.csv"
	grep -E "$KREGEX" "$f" | tail -16 >> "$OUT/${TAG}_pmc_$name.csv"
	echo "$name done" >&2
done
cd "$ROOT"
python3 profiles/kernel_resources.py > "$OUT/${TAG}_kernel_resources.json"
PMC_LOWER_LAUNCHES=2 python3 profiles/pmc_traffic.py "$TAG" "$OUT/${TAG}_pmc_FETCH_SIZE.csv" "$OUT/${TAG}_pmc_WRITE_SIZE.csv" 1000 1000000 4 2 "$OUT/${TAG}_traffic.json" "$OUT/${TAG}_pmc_SQ_INSTS_VALU.csv" > /dev/null
# the bench line of the SAME binary with the measured roofline: the traffic file carries the library's sha256, bench.py refuses another build's
unset PHYAMD_BENCH_BLOCK
python3 bench.py --steps 10 --warmup 2 --traffic-json "$OUT/${TAG}_traffic.json" > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
ls -la "$OUT" >&2
