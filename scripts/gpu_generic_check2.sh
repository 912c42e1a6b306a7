#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-g2}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_$TAG.log
for args in "--config 4 --steps 1000 --warmup 100" "--model wavenet_condition_dsp --streams 512 --steps 500 --warmup 50 --reps 5"; do
  timeout 300 python bench.py $args --no-cpu-baseline --no-side-runs 2>gpurun_out/bench_gen_$TAG.err | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$args', '| value', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'resident', (j.get('resident_launch') or {}).get('value'), 'err', j['max_abs_err_vs_oracle'], 'kernel', j['config']['kernel'])
except Exception as e:
    print('$args', 'FAILED', e)
"
done
bash scripts/gpu_generic_pmc.sh wavenet_a2_max 512 2>&1 | tail -3
