#!/bin/bash
# profiles/quick_pmc.sh TAG "CTR1 CTR2 ..." ["CTRs of a second pass" ...] -- arbitrary counter passes over one evaluation of the
# bench workload (QSQ_ARGS: extra bench.py flags, QSQ_KERNELS: kernel regex); per-kernel sums land in gpurun_out/pmc_TAG/p*.json.
# A diagnostic, not part of the judged figures.  A pass whose counters the profiler rejects is skipped (its .err says why).
set -o pipefail
TAG=${1:?tag}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PHYAMD_BENCH_BLOCK=1000000
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs --no-distinct-check --no-drop-in --steps 1 --warmup 1 $QSQ_ARGS"
KREGEX=${QSQ_KERNELS:-'k_(lower4|upper4)'}
p=0
for ctrs in "$@"; do
  p=$((p+1))
  if rocprofv3 --pmc $ctrs --kernel-trace --output-format csv --kernel-include-regex "$KREGEX" -d /tmp/qpmc_${TAG}_$p -o pmc -- $BENCH > /dev/null 2> $OUT/p$p.err; then
    f=$(find /tmp/qpmc_${TAG}_$p -name 'pmc_counter_collection.csv' | head -1)
    python3 $ROOT/profiles/pmc_sum.py "$f" "$KREGEX" > $OUT/p$p.json
    echo "pass $p ($ctrs) done" >&2
  else
    echo "pass $p ($ctrs) rejected" >&2
  fi
done
