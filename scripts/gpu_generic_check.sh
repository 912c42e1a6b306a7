#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-g1}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_$TAG.log
for args in "--config 4 --steps 20 --warmup 5" "--config 4 --steps 1000 --warmup 100" "--config 4 --steps 300 --warmup 30 --streams 4096 --reps 5" "--config 5 --steps 1000 --warmup 100" "--config 3 --steps 2000 --warmup 200" "--config 3 --steps 20 --warmup 5" "--config 2 --kernel generic --steps 200 --warmup 20 --reps 3" "--model wavenet_condition_dsp --streams 512 --steps 500 --warmup 50 --reps 5"; do
  timeout 300 python bench.py $args --no-cpu-baseline --no-side-runs 2>gpurun_out/bench_gen_$TAG.err | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$args', '| value', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'resident', (j.get('resident_launch') or {}).get('value'), 'err', j['max_abs_err_vs_oracle'], 'kernel', j['config']['kernel'])
except Exception as e:
    print('$args', 'FAILED', e)
"
done
