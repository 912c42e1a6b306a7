#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-l1}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lstm or golden or reference_library" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_$TAG.log
for K in auto; do
  for args in "--steps 20 --warmup 5" "--steps 2000 --warmup 200" "--steps 500 --warmup 50 --fast-tanh 0" "--steps 300 --warmup 30 --streams 8192 --reps 5"; do
    timeout 300 python bench.py --config 3 --kernel $K $args --no-cpu-baseline --no-side-runs 2>gpurun_out/bench_lstm_${K}_$TAG.err | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$K', '$args', '| value', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'resident', (j.get('resident_launch') or {}).get('value'), 'err', j['max_abs_err_vs_oracle'], 'kernel', j['config']['kernel'])
except Exception as e:
    print('$K', '$args', 'FAILED', e)
"
  done
done
