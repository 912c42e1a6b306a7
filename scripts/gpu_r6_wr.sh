#!/bin/bash
# round 6: nam_wn_reg_kernel's dense forms (two wavefronts per SIMD): the GPU suite, then configs 4 and 5 in 500-step regions with
# the dense forms on / off, and what the launches ran as.   TESTS=0: skip the suite
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
if [ "${TESTS:-1}" = "1" ]; then
  ( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
fi
Q() { python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', '$1', j['config']['kernel'], round(j['ms_per_step']*1e3,3), 'us/step', j['value'], 'err', j['max_abs_err_vs_oracle'])
"; }
B="python3 bench.py --gpus 1 --steps 500 --warmup 50 --brief --no-cpu-baseline"
for c in 4 5; do
  NAM_HIP_SESSION_STATS=1 $B --config $c 2>&1 | grep "nam_wn_reg_kernel:" | sort | uniq -c
done
for rep in 1 2; do
for c in 4 5; do
  $B --config $c 2>/dev/null | Q "config $c"
  NAM_HIP_WR_DENSE=0 $B --config $c 2>/dev/null | Q "config $c dense off"
done
done
$B --config 4 --streams 256 2>/dev/null | Q "config 4, 256 streams"
$B --config 4 --streams 1024 2>/dev/null | Q "config 4, 1024 streams"
$B --config 5 --streams 1536 2>/dev/null | Q "config 5, 1536 streams"
