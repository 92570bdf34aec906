#!/bin/bash
# profiles/collect_generic.sh TAG -- rocprofv3 kernel stats of the 20- and 61-state workloads (bench.py --config cfg3 / cfg4),
# run on the GPU box from the repo root; summaries land in gpurun_out/prof_TAG/ (copy what is to be judged into profiles/).
set -eo pipefail
TAG=${1:?tag}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in cfg3 cfg4; do
	rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o kt -- python3 $ROOT/bench.py --no-cpu-baseline --config $cfg --steps 10 --warmup 2 \
		> "$OUT/${TAG}_${cfg}_bench_under_rocprof.json" 2> "$OUT/$cfg.err"
	f=$(find /tmp/prof_$cfg -name 'kt_kernel_stats.csv' | head -1)
	head -1 "$f" > "$OUT/${TAG}_${cfg}_kernel_stats.csv"
	grep -E 'k_lower_gen|k_upper_gen|k_matrix_images|k_transition|k_root|k_reduce_rows|k_scale' "$f" >> "$OUT/${TAG}_${cfg}_kernel_stats.csv" || true
	echo "$cfg done" >&2
done
# MFMA counters of the two hot kernels (own passes, kernel trace only): instruction counts, then matrix-pipe busy cycles
KREGEX='k_(lower_gen|upper_gen)'
for cfg in cfg3 cfg4; do
	pass=0
	for ctrs in "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
		pass=$((pass + 1))
		name=$(echo $ctrs | cut -d' ' -f1)
		rocprofv3 --pmc $ctrs --kernel-trace --output-format csv --kernel-include-regex "$KREGEX" -d /tmp/prof_${cfg}_p$pass -o pmc -- python3 $ROOT/bench.py --no-cpu-baseline \
			--no-other-configs --no-distinct-check --config $cfg --steps 1 --warmup 1 > /dev/null 2> "$OUT/${cfg}_pmc_$name.err"
		f=$(find /tmp/prof_${cfg}_p$pass -name 'pmc_counter_collection.csv' | head -1)
		python3 "$ROOT/profiles/pmc_sum.py" "$f" "$KREGEX" > "$OUT/${TAG}_${cfg}_pmc_$name.json"
		echo "$cfg $name done" >&2
	done
done
cd "$ROOT"
# counter figures per evaluation, tied to this binary: copy to profiles/generic_latest.json for bench.py's other_configs[].mfma
python3 profiles/generic_counters.py "$TAG" "$OUT" > "$OUT/${TAG}_generic_counters.json"
for cfg in cfg3 cfg4; do python3 bench.py --generic-json "$OUT/${TAG}_generic_counters.json" --no-cpu-baseline --config $cfg --steps 10 --warmup 2 > "$OUT/${TAG}_${cfg}_bench.json" 2>> "$OUT/$cfg.err"; done
ls "$OUT" >&2
