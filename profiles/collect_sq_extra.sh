set -eo pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_extra  # copy p*.json into profiles/ as <tag>_walks_sq_pass*.json; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PHYAMD_BENCH_BLOCK=1000000
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --no-distinct-check --steps 1 --warmup 1"
KREGEX='k_(lower4|upper4)'
p=0
for ctrs in "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_IFETCH" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  p=$((p+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv --kernel-include-regex "$KREGEX" -d /tmp/px_$p -o pmc -- $BENCH > /dev/null 2> $OUT/p$p.err
  f=$(find /tmp/px_$p -name 'pmc_counter_collection.csv' | head -1)
  python3 $ROOT/profiles/pmc_sum.py "$f" "$KREGEX" > $OUT/p$p.json
done
