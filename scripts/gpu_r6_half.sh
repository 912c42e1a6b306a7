#!/bin/bash
# round 6: the first buffer of a launch in half units (shipped tree) against variants/ (NAM_AQ_NO_HALF): parity of the shipped tree,
# the in-kernel timeline of a 20-buffer launch, the driver shape and 500-step regions, same box.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
if [ "${TESTS:-1}" = "1" ]; then
  timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_breadth.py -k "bench_shapes and (2 or 22 or 26) or pipelined_kernel or rebase or short_bursts or longer_buffers" 2>&1 | tail -3
  timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_soak.py -k "a1_standard" 2>&1 | tail -2
  timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_tickets.py -k "a1 or standard or ticket or session or persist" 2>&1 | tail -2
fi
cp neuralampmodelercore_amd/lib/libnam_hip.so /tmp/libnam_hip.main.so
Q() { python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', '$1', j['config']['kernel'], round(j['ms_per_step']*1e3,3), 'us/step', j['value'], j.get('region_us'), 'err', j['max_abs_err_vs_oracle'])
"; }
B="python3 bench.py --gpus 1 --no-other-configs --no-side-runs --no-cpu-baseline"
for v in /tmp/libnam_hip.main.so variants/*.so; do
  cp "$v" neuralampmodelercore_amd/lib/libnam_hip.so
  echo "== $(basename $v .so): timeline"
  timeout 120 python tools/a1q_timeline.py 20 2>/dev/null | tail -17
done
for rep in $(seq 1 ${REPS:-2}); do
for v in /tmp/libnam_hip.main.so variants/*.so; do
  cp "$v" neuralampmodelercore_amd/lib/libnam_hip.so
  t=$(basename $v .so)
  timeout 200 $B --steps 20 --warmup 5 2>/dev/null | Q "$t driver"
  timeout 200 $B --steps 500 --warmup 50 --brief 2>/dev/null | Q "$t steady"
done
done
cp /tmp/libnam_hip.main.so neuralampmodelercore_amd/lib/libnam_hip.so
