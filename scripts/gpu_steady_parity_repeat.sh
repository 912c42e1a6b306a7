#!/bin/bash
# hunting an intermittent parity failure of the 500-step regions (max_abs_err 0.562 in 3 of ~14 runs of scripts/gpu_ab_variants.sh)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
B="python3 bench.py --gpus 1 --no-other-configs --no-side-runs --no-cpu-baseline $@"
cp neuralampmodelercore_amd/lib/libnam_hip.so /tmp/libnam_hip.main.so
for i in $(seq 1 ${N:-40}); do
  cp /tmp/libnam_hip.main.so neuralampmodelercore_amd/lib/libnam_hip.so
  timeout 200 $B --steps 20 --warmup 5 > /dev/null 2>&1
  timeout 200 $B --steps 500 --warmup 50 --brief 2> /tmp/err.txt | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  run $i', j['config']['kernel'], round(j['ms_per_step']*1e3,3), 'err', j['max_abs_err_vs_oracle'], j.get('parity_detail'))
"
  grep "PARITY" /tmp/err.txt
done
