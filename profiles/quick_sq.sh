#!/bin/bash
# profiles/quick_sq.sh TAG -- four SQ counter passes over the two 4-state walks of the headline workload (one evaluation each);
# per-kernel sums land in gpurun_out/sq_TAG/p*.json.  A diagnostic for kernel work, not part of the judged roofline figures.
set -eo pipefail
TAG=${1:?tag}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sq_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PHYAMD_BENCH_BLOCK=1000000
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --no-distinct-check --steps 1 --warmup 1 $QSQ_ARGS"  # QSQ_ARGS: extra bench.py flags (e.g. --subst-gradient)
KREGEX=${QSQ_KERNELS:-'k_(lower4|upper4)'}
p=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  p=$((p+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv --kernel-include-regex "$KREGEX" -d /tmp/qsq_${TAG}_$p -o pmc -- $BENCH > /dev/null 2> $OUT/p$p.err
  f=$(find /tmp/qsq_${TAG}_$p -name 'pmc_counter_collection.csv' | head -1)
  python3 $ROOT/profiles/pmc_sum.py "$f" "$KREGEX" > $OUT/p$p.json
  echo "pass $p done" >&2
done
