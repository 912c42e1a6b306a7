#!/bin/bash
# developer tool: A2 on the K-tap kernel, current library vs every library under variants/ on the same box
cd "$GRAFT_REPO_ROOT"
cp neuralampmodelercore_amd/lib/libnam_hip.so /tmp/libnam_hip.orig.so
one() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('  ', j['config']['kernel'], j['config']['launch'], 'streams', j['config']['streams_per_gpu'], 'us/step', round(j['ms_per_step']*1e3,2), 'err', j['max_abs_err_vs_oracle'])
"; }
for v in /tmp/libnam_hip.orig.so variants/*.so; do
  [ -f "$v" ] || continue
  cp "$v" neuralampmodelercore_amd/lib/libnam_hip.so
  echo "== $v"
  one --model A2 --kernel a1_mfma --streams 256 --launch block --steps 400 --warmup 40
  one --model A2 --kernel a1_mfma --streams 256 --launch resident --steps 400 --warmup 40
  one --model A2 --kernel a1_mfma --streams 1024 --launch block --steps 200 --warmup 20
done
cp /tmp/libnam_hip.orig.so neuralampmodelercore_amd/lib/libnam_hip.so
