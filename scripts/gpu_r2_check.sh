#!/bin/bash
# Runs on the GPU box (via gpurun): the -m gpu parity suite, then bench.py on every BASELINE config (driver-shaped
# short run + a long run). Usage: gpurun -- 'bash scripts/gpu_r2_check.sh TAG [configs...]'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r2a}; shift
CONFIGS=${@:-2 3 4 5}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_$TAG.log
for c in $CONFIGS; do
  # what the driver runs (20 steps / 5 warm-up), then the long form
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_c${c}_short_$TAG.json 2> gpurun_out/bench_c${c}_short_$TAG.err
  timeout 900 python bench.py --config $c > gpurun_out/bench_c${c}_$TAG.json 2> gpurun_out/bench_c${c}_$TAG.err
  python - <<PY
import json
for f in ("gpurun_out/bench_c${c}_short_$TAG.json", "gpurun_out/bench_c${c}_$TAG.json"):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("config $c", f.split("/")[-1], "value", j["value"], "us/step", round(j["ms_per_step"] * 1e3, 2), "kernel", j["config"]["kernel"],
              "frac", j["roofline"]["frac"], "resident", (j.get("resident_launch") or {}).get("value"), "lat", j.get("latency_us"),
              "err", j["max_abs_err_vs_oracle"], "ft_off", (j.get("fast_tanh_off") or {}).get("value"), "zeros", (j.get("zeros_input") or {}).get("value"))
    except Exception as e:
        print("config $c", f, "FAILED", e)
PY
done
