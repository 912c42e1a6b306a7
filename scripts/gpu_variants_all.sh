#!/bin/bash
# developer tool: every library under variants/ against the current one on the headline, LSTM and A2 shapes
cd "$GRAFT_REPO_ROOT"
cp neuralampmodelercore_amd/lib/libnam_hip.so /tmp/libnam_hip.orig.so
one() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('  ', j['config']['workload'].split(',')[0], j['config']['kernel'], j['config']['launch'], 'streams', j['config']['streams_per_gpu'], 'us/step', round(j['ms_per_step']*1e3,2), 'err', j['max_abs_err_vs_oracle'])
"; }
for v in /tmp/libnam_hip.orig.so variants/*.so; do
  [ -f "$v" ] || continue
  cp "$v" neuralampmodelercore_amd/lib/libnam_hip.so
  echo "== $v"
  one --launch block --steps 1500 --warmup 150
  one --launch resident --steps 1500 --warmup 150
  one --launch resident --streams 4096 --steps 200 --warmup 20
  one --model lstm --streams 1024 --launch block --steps 400 --warmup 40
  one --model A2 --kernel a1_mfma --streams 256 --launch block --steps 300 --warmup 30
  one --model A2 --kernel a1_mfma --streams 256 --launch resident --steps 300 --warmup 30
  one --model A2 --kernel a1_mfma --streams 2048 --launch resident --steps 200 --warmup 20
done
cp /tmp/libnam_hip.orig.so neuralampmodelercore_amd/lib/libnam_hip.so
