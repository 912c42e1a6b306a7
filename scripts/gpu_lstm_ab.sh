#!/bin/bash
# developer tool: LSTM fixtures, current library vs every library under variants/ on the same box
cd "$GRAFT_REPO_ROOT"
cp neuralampmodelercore_amd/lib/libnam_hip.so /tmp/libnam_hip.orig.so
for v in /tmp/libnam_hip.orig.so variants/*.so; do
  [ -f "$v" ] || continue
  cp "$v" neuralampmodelercore_amd/lib/libnam_hip.so
  echo "== $v"
  for m in lstm synth_lstm_h18x2 synth_lstm_h10x2; do
    for l in block resident; do
      python bench.py --model $m --streams 1024 --launch $l --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('  $m $l', 'xRT', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'err', j['max_abs_err_vs_oracle'])
"
    done
  done
done
cp /tmp/libnam_hip.orig.so neuralampmodelercore_amd/lib/libnam_hip.so
