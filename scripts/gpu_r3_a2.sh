#!/bin/bash
# round 3: nam_kp_kernel (the A2 topology as a pipeline of wave sets; NAM_HIP_NO_PIPE=1: nam_kt_mfma_kernel): parity, then A2-Full at 256 streams
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 900 python -m pytest tests/test_gpu_breadth.py tests/test_gpu_parity.py tests/test_container.py -m gpu -q --timeout=300 -p no:cacheprovider -k "a2 or A2 or container or bench_shapes" > gpurun_out/r3_a2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_a2_tests.log
tail -15 gpurun_out/r3_a2_tests.log | cut -c1-300
{
for np_ in 0 1; do
  for shape in "--steps 20 --warmup 5" "--steps 1000 --warmup 100"; do
    NAM_HIP_NO_PIPE=$np_ timeout 300 python bench.py --model A2 --streams 256 $shape --no-other-configs --no-cpu-baseline --no-side-runs 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print('A2-Full NO_PIPE=$np_', j['config']['kernel'], 'persistent', j['config']['persistent_block_mode'], 'steps', j['steps'], 'us/step', round(j['ms_per_step'] * 1e3, 2), 'xRT', j['value'], 'err', j['max_abs_err_vs_oracle'], 'resident', (j.get('resident_launch') or {}).get('value'))
"
  done
done
} 2>&1 | tee gpurun_out/r3_a2.txt
