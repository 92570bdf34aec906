#!/bin/bash
# A/B of runtime switches on one box: profiles/ab.sh "VAR=0" "VAR=1" ...  (each argument: env assignments for one bench run; "-" = none)
# prints ms per step and the two passes' ms per evaluation of the headline workload (bench.py defaults, no side measurements)
mkdir -p gpurun_out/ab
for spec in "$@"; do
	envs=""
	[ "$spec" != "-" ] && envs="$spec"
	out=gpurun_out/ab/$(echo "$spec" | tr ' =/' '___').json
	env $envs python bench.py --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline --no-other-configs --no-drop-in --no-distinct-check ${AB_ARGS} > "$out" 2> "$out.err" || { echo "FAILED: $spec"; tail -5 "$out.err"; exit 1; }
	python - "$out" "$spec" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
r = d.get("roofline", {})
m = r.get("ms_per_eval", {})
print(f"{sys.argv[2]:40s} value {d['value']:.2f}  ms/step {d['ms_per_step']:.2f}  lower {m.get('lower_ms')}  upper {m.get('upper_ms')}")
PY
done
