#!/bin/bash
# developer tool: VALU / generic kernel shapes, current library vs every library under variants/ on the same box
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
  one --model wavenet_a2_max --streams 512 --launch block --steps 200 --warmup 20
  one --model A2 --kernel a1 --streams 256 --launch block --steps 200 --warmup 20
  one --model A2 --kernel a1 --streams 2048 --launch block --steps 100 --warmup 10
  one --model A2 --kernel a1 --streams 4096 --launch resident --steps 50 --warmup 5
  one --model slimmable_wavenet --streams 768 --launch block --steps 300 --warmup 30
  one --model wavenet_a1_standard --kernel a1 --streams 4096 --launch resident --steps 50 --warmup 5
done
cp /tmp/libnam_hip.orig.so neuralampmodelercore_amd/lib/libnam_hip.so
