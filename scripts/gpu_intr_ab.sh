#!/bin/bash
# A/B: completion signals polled (HSA_ENABLE_INTERRUPT=0) vs interrupt-driven, driver-shaped and default runs
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for mode in default poll default poll; do
  for c in 2 5; do
    if [ $mode = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
    timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-side-runs > gpurun_out/ab_${mode}_c$c.json 2> gpurun_out/ab_${mode}_c$c.err
    python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/ab_${mode}_c$c.json").read().strip().splitlines()[-1])
    print("$mode config $c value", j["value"], "region_us", j.get("region_us"))
except Exception as e:
    print("$mode config $c FAILED", e); print(open("gpurun_out/ab_${mode}_c$c.err").read()[-800:])
PY
  done
done
unset HSA_ENABLE_INTERRUPT
timeout 600 python bench.py --config 2 --no-cpu-baseline --no-side-runs | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default-steps interrupt', j['value'])"
HSA_ENABLE_INTERRUPT=0 timeout 600 python bench.py --config 2 --no-cpu-baseline --no-side-runs | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default-steps poll', j['value'])"
