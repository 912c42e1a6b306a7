#!/bin/bash
# developer tool: bench every prebuilt library under variants/ (built with different -D knobs) on the headline shapes
cd "$GRAFT_REPO_ROOT"
cp neuralampmodelercore_amd/lib/libnam_hip.so /tmp/libnam_hip.orig.so
for v in /tmp/libnam_hip.orig.so variants/*.so; do
  cp "$v" neuralampmodelercore_amd/lib/libnam_hip.so
  echo "== $v"
  for args in "--launch block --steps 2000 --warmup 200" "--launch resident --steps 2000 --warmup 200" "--launch resident --streams 4096 --steps 300 --warmup 30"; do
    python bench.py --kernel ${K:-a1_mfma} $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(' ', j['config']['launch'], 'streams', j['config']['streams_per_gpu'], 'us/step', round(j['ms_per_step']*1e3,2), 'err', j['max_abs_err_vs_oracle'])
"
  done
done
cp /tmp/libnam_hip.orig.so neuralampmodelercore_amd/lib/libnam_hip.so
