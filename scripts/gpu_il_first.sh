#!/bin/bash
# first contact of nam_a1_il_kernel: parity, then A/B against the wave-specialised kernel
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-il1}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "interleaved" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_$TAG.log
for K in a1_il a1_mfma; do
  for args in "--steps 20 --warmup 5" "--steps 2000 --warmup 200" "--launch resident --steps 2000 --warmup 200 --reps 5" "--launch resident --streams 4096 --steps 300 --warmup 30 --reps 5" "--streams 1024 --steps 500 --warmup 50 --reps 5"; do
    timeout 300 python bench.py --kernel $K $args --no-cpu-baseline --no-side-runs 2>gpurun_out/bench_${K}_$TAG.err | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$K', '$args', '| value', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'frac', j['roofline']['frac'], 'resident', (j.get('resident_launch') or {}).get('value'), 'err', j['max_abs_err_vs_oracle'], 'kernel', j['config']['kernel'])
except Exception as e:
    print('$K', '$args', 'FAILED', e)
"
  done
done
