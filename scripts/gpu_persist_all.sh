#!/bin/bash
# persistent block mode on the three session kernels: parity tests, then configs 2, 3, 4 driver-shaped and long
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-pa1}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "persistent or lstm or register_resident" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_$TAG.log
for c in 3 4 2; do
 for args in "--steps 20 --warmup 5" "" "--persistent 0"; do
  timeout 600 python bench.py --config $c $args --no-cpu-baseline --no-side-runs 2>gpurun_out/bench_$TAG.err | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('config $c $args', '| value', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'resident', (j.get('resident_launch') or {}).get('value'), 'err', j['max_abs_err_vs_oracle'], 'kernel', j['config']['kernel'], 'persist', j['config']['persistent_block_mode'], 'region', j.get('region_us'))
except Exception as e:
    print('config $c $args', 'FAILED', e)
"
  tail -2 gpurun_out/bench_$TAG.err | cut -c1-300
 done
done
