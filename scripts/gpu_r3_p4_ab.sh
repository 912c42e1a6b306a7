#!/bin/bash
# round 3: same-box A/B of the headline kernels: p2 / p3 / p4 (stage counts, queue forms); parity tests first
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=5000
timeout 900 python -m pytest tests/test_gpu_breadth.py tests/test_gpu_parity.py -k "pipelined or bench_shapes or interleaved or persistent or headline or prewarm_cache or long_render or long_resident" -m gpu -q --timeout=180 -p no:cacheprovider > gpurun_out/r3_p4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_p4_tests.log
tail -5 gpurun_out/r3_p4_tests.log
cp neuralampmodelercore_amd/lib/libnam_hip.so /tmp/libnam_hip.orig.so
bench_line() {
  for shape in "--steps 20 --warmup 5" "--steps 2000 --warmup 200"; do
  timeout 300 python bench.py $shape --no-other-configs --no-cpu-baseline --no-side-runs 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print('   ', j['config']['kernel'], 'steps', j['steps'], 'us/step', round(j['ms_per_step'] * 1e3, 3), 'xRT', j['value'], 'err', j['max_abs_err_vs_oracle'])
"
  done
}
{
echo "== p2"; NAM_HIP_NO_PIPE=1 bench_line
for v in /tmp/libnam_hip.orig.so variants/libnam_hip_*.so; do
  cp "$v" neuralampmodelercore_amd/lib/libnam_hip.so
  echo "== p4 $v"
  bench_line
done
echo "== p2 again"; NAM_HIP_NO_PIPE=1 bench_line
} 2>&1 | tee gpurun_out/r3_p4_ab.log
cp /tmp/libnam_hip.orig.so neuralampmodelercore_amd/lib/libnam_hip.so
