#!/bin/bash
# the LSTM kernels: their tests, then config 3 of the bench (500-step regions) and the 2 x 18 fixture
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 900 python -m pytest tests -m gpu -q -x --timeout=300 -p no:cacheprovider -k "lstm or LSTM" > gpurun_out/r4_lstm_tests.log 2>&1
echo "lstm tests rc=$? $(tail -1 gpurun_out/r4_lstm_tests.log)"; grep "^FAILED\|^ERROR" gpurun_out/r4_lstm_tests.log | head
for rep in 1 2; do
timeout 600 python3 bench.py --config 3 --gpus 1 --steps 500 --warmup 50 --brief --no-cpu-baseline 2> gpurun_out/r4_bench_c3.err | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('config 3', j['steps'], 'steps:', j['value'], 'xRT', round(j['ms_per_step']*1e3, 3), 'us/step', j['config']['kernel'], 'err', j['max_abs_err_vs_oracle'])
"
done
timeout 600 python3 bench.py --model synth_lstm_h18x2 --streams 1024 --gpus 1 --steps 500 --warmup 50 --brief --no-cpu-baseline 2>> gpurun_out/r4_bench_c3.err | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('2x18', j['steps'], 'steps:', j['value'], 'xRT', round(j['ms_per_step']*1e3, 3), 'us/step', j['config']['kernel'], 'err', j['max_abs_err_vs_oracle'])
"
