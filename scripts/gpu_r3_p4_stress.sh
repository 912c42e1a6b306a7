#!/bin/bash
# round 3: nam_a1_p4_kernel — parity tests, then a long persistent soak per stage count with EVERY stream checked at the end
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=5000
timeout 900 python -m pytest tests/test_gpu_breadth.py tests/test_gpu_parity.py -k "pipelined or bench_shapes or interleaved or persistent or headline or prewarm_cache or long_render or long_resident" -m gpu -q -x --timeout=180 -p no:cacheprovider > gpurun_out/r3_p4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_p4_tests.log
tail -5 gpurun_out/r3_p4_tests.log
cp neuralampmodelercore_amd/lib/libnam_hip.so /tmp/libnam_hip.orig.so
for v in /tmp/libnam_hip.orig.so variants/libnam_hip_p4s*.so; do
  cp "$v" neuralampmodelercore_amd/lib/libnam_hip.so
  echo "== $v"
  timeout 600 python tools/persist_soak.py 256 3000 3
  timeout 300 python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-side-runs 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print('   ', j['config']['kernel'], 'steps', j['steps'], 'us/step', round(j['ms_per_step'] * 1e3, 3), 'xRT', j['value'], 'err', j['max_abs_err_vs_oracle'])
"
  timeout 300 python bench.py --steps 2000 --warmup 200 --no-other-configs --no-cpu-baseline --no-side-runs 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print('   ', j['config']['kernel'], 'steps', j['steps'], 'us/step', round(j['ms_per_step'] * 1e3, 3), 'xRT', j['value'], 'err', j['max_abs_err_vs_oracle'])
"
done 2>&1 | tee gpurun_out/r3_p4_stress.log
cp /tmp/libnam_hip.orig.so neuralampmodelercore_amd/lib/libnam_hip.so
