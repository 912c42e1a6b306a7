#!/bin/bash
# round 3: nam_wn_reg_kernel with two wavefronts per stream (the op program cut in two): parity, then configs 4 / 5 and a few others
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x > gpurun_out/r3_wr2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_wr2_tests.log
tail -8 gpurun_out/r3_wr2_tests.log
{
for np_ in 0 1; do
  for c in 4 5; do
    timeout 300 python bench.py --config $c --steps 500 --warmup 50 --no-other-configs --no-cpu-baseline --no-side-runs 2>/dev/null | grep '^{' | NP=$np_ python -c "
import sys, json, os
for l in sys.stdin:
    j = json.loads(l)
    print('config', j['config']['baseline_config'], 'NO_PIPE', os.environ['NP'], j['config']['kernel'], 'us/step', round(j['ms_per_step'] * 1e3, 2), 'xRT', j['value'], 'err', j['max_abs_err_vs_oracle'], 'resident', (j.get('resident_launch') or {}).get('value'))
" 
  done
  export NAM_HIP_NO_PIPE=1
done
unset NAM_HIP_NO_PIPE
for m in wavenet_condition_dsp synth_multich; do
  for np_ in 0 1; do
    NAM_HIP_NO_PIPE=$np_ timeout 300 python bench.py --model $m --streams 256 --steps 500 --warmup 50 --no-other-configs --no-cpu-baseline --no-side-runs 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print('$m 256 streams NO_PIPE=$np_', j['config']['kernel'], 'us/step', round(j['ms_per_step'] * 1e3, 2), 'xRT', j['value'], 'err', j['max_abs_err_vs_oracle'])
"
  done
done
} 2>&1 | tee gpurun_out/r3_wr2.txt
