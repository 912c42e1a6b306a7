#!/bin/bash
# round 3: nam_a1_p3_kernel (two wave sets per stream) — parity first, then A/B against nam_a1_p2_kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=5000
timeout 900 python -m pytest "tests/test_gpu_breadth.py::test_two_wave_set_kernel_equals_the_four_wave_kernel" "tests/test_gpu_breadth.py::test_bench_shapes_every_stream_in_persistent_mode" tests/test_gpu_parity.py -k "two_wave or bench_shapes or interleaved or persistent or headline or prewarm_cache or long_render or long_resident" -m gpu -q -x --timeout=180 -p no:cacheprovider > gpurun_out/r3_p3_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_p3_tests.log
tail -25 gpurun_out/r3_p3_tests.log
for np3 in 1 0; do
  for shape in "--steps 20 --warmup 5" "--steps 2000 --warmup 200"; do
    echo "== NAM_HIP_NO_P3=$np3 $shape" >> gpurun_out/r3_p3_bench.log
    NAM_HIP_NO_P3=$np3 timeout 300 python bench.py $shape --no-other-configs --no-cpu-baseline --no-side-runs 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print(json.dumps({k: j[k] for k in ('value', 'ms_per_step', 'max_abs_err_vs_oracle', 'region_us', 'resident_launch')} | {'kernel': j['config']['kernel'], 'frac': j['roofline']['frac']}))
" >> gpurun_out/r3_p3_bench.log
  done
done
cat gpurun_out/r3_p3_bench.log
