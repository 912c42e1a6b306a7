#!/bin/bash
# round 3: sessions with more one-workgroup-per-stream workgroups than CUs (they take turns)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 900 python -m pytest tests/test_gpu_breadth.py -m gpu -q --timeout=300 -p no:cacheprovider -k "bench_shapes" > gpurun_out/r3_turns_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_turns_tests.log
tail -6 gpurun_out/r3_turns_tests.log
{
for streams in 256 512 1024 2048; do
  for pers in 1 0; do
    timeout 300 python bench.py --config 2 --streams $streams --persistent $pers --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-side-runs 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print('a1_standard streams $streams persistent $pers', j['config']['kernel'], j['config']['persistent_block_mode'], 'us/step', round(j['ms_per_step'] * 1e3, 2), 'xRT', j['value'], 'err', j['max_abs_err_vs_oracle'])
"
  done
done
for streams in 512 1024; do
  for pers in 1 0; do
    timeout 300 python bench.py --model A2 --streams $streams --persistent $pers --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-side-runs 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print('A2-Full streams $streams persistent $pers', j['config']['kernel'], j['config']['persistent_block_mode'], 'us/step', round(j['ms_per_step'] * 1e3, 2), 'xRT', j['value'], 'err', j['max_abs_err_vs_oracle'])
"
  done
done
} 2>&1 | tee gpurun_out/r3_turns.txt
