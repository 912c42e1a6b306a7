#!/bin/bash
# Round-end set: the whole -m gpu suite, the driver-shaped and default bench lines of configs 2..5 (JSON kept), and a
# rocprofv3 kernel trace of the DEFAULT bench command of configs 2..5 (persistent block mode: few long dispatches).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-fin}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_$TAG.log
for c in ${CONFIGS:-2 3 4 5}; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 > gpurun_out/bench_c${c}_driver_$TAG.json 2> gpurun_out/bench_c${c}_driver_$TAG.err
  timeout 900 python bench.py --config $c > gpurun_out/bench_c${c}_$TAG.json 2> gpurun_out/bench_c${c}_$TAG.err
  timeout 900 python bench.py --config $c --persistent 0 --no-side-runs --no-cpu-baseline > gpurun_out/bench_c${c}_np_$TAG.json 2> gpurun_out/bench_c${c}_np_$TAG.err
  python - <<PY
import json
for f in ("gpurun_out/bench_c${c}_driver_$TAG.json", "gpurun_out/bench_c${c}_$TAG.json", "gpurun_out/bench_c${c}_np_$TAG.json"):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("config $c", f.split("/")[-1], "value", j["value"], "us/step", round(j["ms_per_step"] * 1e3, 2), "kernel", j["config"]["kernel"], "persist", j["config"].get("persistent_block_mode"),
              "frac", j["roofline"]["frac"], "resident", (j.get("resident_launch") or {}).get("value"), "lat_p50", (j.get("latency_us") or {}).get("p50"),
              "err", j["max_abs_err_vs_oracle"], "cpu", (j.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print("config $c", f, "FAILED", e)
PY
done
for c in ${CONFIGS:-2 3 4 5}; do
  D=gpurun_out/ptrace_c${c}_$TAG
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python bench.py --config $c --no-side-runs --no-cpu-baseline --reps 3 > gpurun_out/ptrace_bench_c${c}_$TAG.json 2> gpurun_out/ptrace_c${c}_$TAG.err
  find $D -name "trace_kernel_stats.csv" -exec cp {} gpurun_out/persistent_kernel_stats_c${c}_$TAG.csv \;
  head -3 gpurun_out/persistent_kernel_stats_c${c}_$TAG.csv | cut -c1-200
done
