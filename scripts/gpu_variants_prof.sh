#!/bin/bash
# developer tool: barrier/segment profile + resident bench for prebuilt library variants
cd "$GRAFT_REPO_ROOT"
L=neuralampmodelercore_amd/lib
cp $L/libnam_hip.so $L/libnam_hip.so.base
for v in base "$@"; do
  cp $L/libnam_hip.so.$v $L/libnam_hip.so
  echo "== $v"
  python bench.py --kernel a1_mfma --launch resident --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('resident us/step', round(j['ms_per_step']*1e3,2))
"
  timeout 100 python tools/mfma_barrier_profile.py 256 3200 | sed -n '1,2p;9p'
done
cp $L/libnam_hip.so.base $L/libnam_hip.so
