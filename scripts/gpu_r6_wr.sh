#!/bin/bash
# round 6: configs 4 and 5 (and anything else given) in 500-step regions: dense forms on / off
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
Q() { python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', '$1', j['config']['kernel'], round(j['ms_per_step']*1e3,3), 'us/step', j['value'], 'err', j['max_abs_err_vs_oracle'])
"; }
B="python3 bench.py --gpus 1 --steps 500 --warmup 50 --brief --no-cpu-baseline"
for rep in 1 2; do
for c in 4 5; do
  $B --config $c 2>/dev/null | Q "config $c"
  NAM_HIP_WR_DENSE=0 $B --config $c 2>/dev/null | Q "config $c dense off"
done
done
$B --config 4 --streams 256 2>/dev/null | Q "config 4, 256 streams"
$B --config 5 --streams 1536 2>/dev/null | Q "config 5, 1536 streams"
