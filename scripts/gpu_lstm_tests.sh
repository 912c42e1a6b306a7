#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lstm" > gpurun_out/pytest_lstm2.log 2>&1; echo "lstm tests rc=$?"; tail -15 gpurun_out/pytest_lstm2.log
for m in synth_lstm_h32 synth_lstm_h24x2io; do
timeout 600 python bench.py --config 3 --model $m --no-cpu-baseline --no-side-runs > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err
python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/bench_$m.json").read().strip().splitlines()[-1])
    print("$m", "value", j["value"], "us/step", round(j["ms_per_step"] * 1e3, 2), "kernel", j["config"]["kernel"], "persist", j["config"].get("persistent_block_mode"), "err", j["max_abs_err_vs_oracle"])
except Exception as e:
    print("$m FAILED", e); print(open("gpurun_out/bench_$m.err").read()[-1200:])
PY
done
