#!/bin/bash
# round 3: nam_wn_reg_kernel with two / four wavefronts per stream (the op program cut up): parity (whole suite), then
# config 4 and a few 256-stream cases with 1 / 2 / 4 waves per stream
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x > gpurun_out/r3_wr2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_wr2_tests.log
tail -8 gpurun_out/r3_wr2_tests.log
{
for st in 4 2 1; do
  export NAM_HIP_WR_STAGES=$st
  for c in 4 5; do
    timeout 300 python bench.py --config $c --steps 500 --warmup 50 --no-other-configs --no-cpu-baseline --no-side-runs 2>/dev/null | grep '^{' | python -c "
import sys, json, os
for l in sys.stdin:
    j = json.loads(l)
    print('config', j['config']['baseline_config'], 'max stages', os.environ['NAM_HIP_WR_STAGES'], j['config']['kernel'], 'us/step', round(j['ms_per_step'] * 1e3, 2), 'xRT', j['value'], 'err', j['max_abs_err_vs_oracle'])
"
  done
  for m in wavenet_condition_dsp synth_multich synth_a1_nano wavenet_a2_max; do
    timeout 300 python bench.py --model $m --streams 256 --steps 500 --warmup 50 --no-other-configs --no-cpu-baseline --no-side-runs 2>/dev/null | grep '^{' | M=$m python -c "
import sys, json, os
for l in sys.stdin:
    j = json.loads(l)
    print(os.environ['M'], '256 streams, max stages', os.environ['NAM_HIP_WR_STAGES'], j['config']['kernel'], 'us/step', round(j['ms_per_step'] * 1e3, 2), 'xRT', j['value'], 'err', j['max_abs_err_vs_oracle'])
"
  done
done
} 2>&1 | tee gpurun_out/r3_wr2.txt
