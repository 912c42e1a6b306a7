#!/bin/bash
# round 3: nam_kp_kernel (the A2 topology as a pipeline of wave sets): parity, then the default bench line (A2-Full rides in other_configs)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 900 python -m pytest tests/test_gpu_breadth.py tests/test_gpu_parity.py tests/test_container.py -m gpu -q --timeout=300 -p no:cacheprovider -k "a2 or A2 or container" > gpurun_out/r3_a2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_a2_tests.log
tail -15 gpurun_out/r3_a2_tests.log
SECONDS=0; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
echo "bench wall ${SECONDS} s"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r3_bench_default.json"))
print("config 2", j["value"], j["ms_per_step"])
for k, v in j["other_configs"].items():
    print(k, v.get("value"), v.get("ms_per_step"), (v.get("config") or {}).get("kernel"), v.get("max_abs_err_vs_oracle"), v.get("run_s"), v.get("error"))
PY
