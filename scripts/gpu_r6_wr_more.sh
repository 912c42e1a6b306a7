#!/bin/bash
# round 6: where do nam_wn_reg_kernel's dense forms pay? config 4's model at 256 .. 2048 streams, dense on / off
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
B="python3 bench.py --gpus 1 --steps 500 --warmup 50 --brief --no-cpu-baseline --config 4"
for streams in 256 512 768 1024 1536 2048; do
for env in "X=1" "NAM_HIP_WR_DENSE=0"; do
  echo "== $streams streams, $env"
  env $env NAM_HIP_SESSION_STATS=1 $B --streams $streams 2>&1 | grep -E "nam_wn_reg_kernel:|^\{" | python -c "
import sys, json
seen = set()
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('  ', round(j['ms_per_step']*1e3,3), 'us/step', j['value'], j['max_abs_err_vs_oracle'])
    elif l not in seen:
        seen.add(l); print('  ', l.strip())
"
done
done
