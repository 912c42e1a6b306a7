#!/bin/bash
# driver-shaped run (--steps 20 --warmup 5) of the headline config, persistent block mode: spin-up on/off, and a
# kernel-trace + HIP-API timeline of the same run
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for extra in "--spinup-ms 150" "--spinup-ms 0" "--spinup-ms 150" "--spinup-ms 400"; do
  timeout 300 python bench.py --steps 20 --warmup 5 $extra --no-cpu-baseline --no-side-runs 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$extra', '| value', j['value'], 'region', j.get('region_us'), 'all', j['repetitions']['ms_per_step_all'])
"
done
