#!/bin/bash
# round 3: the whole GPU suite, then the driver's bench command (with other_configs) and the default one
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/r3_full_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_full_tests.log
tail -15 gpurun_out/r3_full_tests.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r3_bench_driver.json 2> gpurun_out/r3_bench_driver.err
tail -3 gpurun_out/r3_bench_driver.err
python - <<'PY'
import json
for l in open("gpurun_out/r3_bench_driver.json"):
    if l.startswith("{"):
        j = json.loads(l)
        print("driver:", j["value"], j["ms_per_step"], j["config"]["kernel"], "roofline", {k: j["roofline"].get(k) for k in ("frac", "kernel_time_frac", "counter_frac", "floor_us")}, "cpu", j.get("cpu_baseline", {}).get("value"), j.get("cpu_baseline", {}).get("O2"))
        for c, r in (j.get("other_configs") or {}).items():
            print("  config", c, {k: r.get(k) for k in ("value", "ms_per_step", "max_abs_err_vs_oracle", "run_s", "error")}, (r.get("roofline") or {}).get("frac"), (r.get("cpu_baseline") or {}).get("value"))
PY
