#!/bin/bash
# round 6: config 4's two-wave cut (NAM_HIP_WR_CUT2 = the op the second wave starts at; default: wr_program_cuts' choice)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
B="python3 bench.py --gpus 1 --steps 500 --warmup 50 --brief --no-cpu-baseline --config 4"
for rep in 1 2; do
for k in default 5 6 8; do
  if [ $k = default ]; then unset NAM_HIP_WR_CUT2; else export NAM_HIP_WR_CUT2=$k; fi
  $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   cut $k', round(j['ms_per_step']*1e3,3), 'us/step', j['value'], 'err', j['max_abs_err_vs_oracle'])
"
done
done
