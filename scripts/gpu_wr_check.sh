#!/bin/bash
# nam_wn_reg_kernel: parity tests, then config 4 (wavenet_a2_max, 512 streams) driver-shaped and long, AUTO vs the op interpreter
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-wr1}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "register_resident or wavenet_matches_oracle or multichannel" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_$TAG.log
for args in "--config 4 --steps 20 --warmup 5" "--config 4" "--config 4 --kernel generic --steps 200 --warmup 20"; do
  timeout 600 python bench.py $args --no-cpu-baseline 2>gpurun_out/bench_$TAG.err | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$args', '| value', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'resident', (j.get('resident_launch') or {}).get('value'), 'err', j['max_abs_err_vs_oracle'], 'kernel', j['config']['kernel'], 'lat', (j.get('latency_us') or {}).get('p50'))
except Exception as e:
    print('$args', 'FAILED', e)
"
  tail -3 gpurun_out/bench_$TAG.err | cut -c1-300
done
