#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-ps1}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "persistent or interleaved or lstm or register_resident" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_$TAG.log
for args in "--persistent 1 --steps 20 --warmup 5" "--persistent 1 --steps 2000 --warmup 200" "--persistent 0 --steps 2000 --warmup 200" "--persistent 1 --kernel a1_il --steps 2000 --warmup 200" "--persistent 0 --kernel a1_il --steps 2000 --warmup 200"; do
  timeout 300 python bench.py $args --no-cpu-baseline 2>gpurun_out/bench_ps_$TAG.err | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$args', '| value', j['value'], 'region', j.get('region_us'), 'us/step', round(j['ms_per_step']*1e3,2), 'frac', j['roofline']['frac'], 'resident', (j.get('resident_launch') or {}).get('value'), 'err', j['max_abs_err_vs_oracle'], 'kernel', j['config']['kernel'], 'persist', j['config']['persistent_block_mode'], 'lat', (j.get('latency_us') or {}).get('p50'), 'enq', j['host_enqueue_us_per_step'])
except Exception as e:
    print('$args', 'FAILED', e)
"
  tail -3 gpurun_out/bench_ps_$TAG.err | cut -c1-300
done
