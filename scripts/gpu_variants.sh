#!/bin/bash
# developer tool: bench several prebuilt library variants (lib/libnam_hip.so.<tag>) back to back
cd "$GRAFT_REPO_ROOT"
L=neuralampmodelercore_amd/lib
cp $L/libnam_hip.so $L/libnam_hip.so.base
for v in base "$@"; do
  cp $L/libnam_hip.so.$v $L/libnam_hip.so
  for args in "--launch block --steps 3000 --warmup 300"; do
    python bench.py --kernel ${KERNEL:-a1_mfma} $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$v', j['config']['launch'], 'us/step', round(j['ms_per_step']*1e3,2), 'err', j['max_abs_err_vs_oracle'])
"
  done
done
cp $L/libnam_hip.so.base $L/libnam_hip.so
