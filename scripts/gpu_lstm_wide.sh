#!/bin/bash
# nam_lstm_wide_kernel: the LSTM / persistent tests, then the 2 x 18 fixture at 1,024 streams (persistent, one launch per
# step, the matrix-core kernel) and config 3 as the regression check.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-lw}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lstm or persistent_block_mode_matches or container" > gpurun_out/pytest_lstm_$TAG.log 2>&1; echo "lstm tests rc=$?"; tail -5 gpurun_out/pytest_lstm_$TAG.log
run() { # name, extra args
  timeout 600 python bench.py --config 3 --no-cpu-baseline --no-side-runs $2 > gpurun_out/bench_lstm_$1_$TAG.json 2> gpurun_out/bench_lstm_$1_$TAG.err
  python - <<PY
import json
f = "gpurun_out/bench_lstm_$1_$TAG.json"
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print("$1", "value", j["value"], "us/step", round(j["ms_per_step"] * 1e3, 2), "kernel", j["config"]["kernel"], "persist", j["config"].get("persistent_block_mode"),
          "resident", (j.get("resident_launch") or {}).get("value"), "err", j["max_abs_err_vs_oracle"])
except Exception as e:
    print("$1 FAILED", e); print(open(f.replace(".json", ".err")).read()[-1200:])
PY
}
run h18x2 "--model synth_lstm_h18x2"
run h18x2_np "--model synth_lstm_h18x2 --persistent 0"
run h18x2_mfma "--model synth_lstm_h18x2 --kernel a1_mfma"
run h10x2 "--model synth_lstm_h10x2"
run lstm ""
