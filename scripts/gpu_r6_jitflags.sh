#!/bin/bash
# round 6: A/B of kernel_wn_reg.hip knobs through the per-model compile (NAM_HIP_JIT_FLAGS), config 4's model at 256 / 512 / 1,024 streams
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
B="python3 bench.py --gpus 1 --steps 500 --warmup 50 --brief --no-cpu-baseline --config 4"
for rep in 1 2; do
for flags in "" "$@"; do
  for streams in 256 512 1024; do
    NAM_HIP_JIT_FLAGS="$flags" $B --streams $streams 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   [$flags]', $streams, 'streams', round(j['ms_per_step']*1e3,3), 'us/step', 'err', j['max_abs_err_vs_oracle'])
"
  done
done
done
