#!/bin/bash
# round 4: A/B of nam_a1_q_kernel build knobs (variants/libnam_hip_aq_*.so), same box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cp neuralampmodelercore_amd/lib/libnam_hip.so /tmp/libnam_hip.orig.so
for v in variants/libnam_hip_aq_*.so; do
  cp "$v" neuralampmodelercore_amd/lib/libnam_hip.so
  echo "== $v"
  timeout 200 python tools/a1q_time.py 2>&1 | grep -v amdgpu.ids
  timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('   driver shape', j['config']['kernel'], round(j['ms_per_step']*1e3,3), 'us/step', j['region_us'], 'err', j['max_abs_err_vs_oracle'])
"
done 2>&1 | tee gpurun_out/r4_variants.txt
cp /tmp/libnam_hip.orig.so neuralampmodelercore_amd/lib/libnam_hip.so
