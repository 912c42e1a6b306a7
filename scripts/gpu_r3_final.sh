#!/bin/bash
# round 3, last check of the tree as shipped: the whole GPU suite (SUITE_REPS times, default once), the soak of every pipelined kernel, smoke, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
for i in $(seq 1 ${SUITE_REPS:-1}); do
  timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/r3_final_tests_$i.log 2>&1
  echo "suite run $i rc=$? $(tail -1 gpurun_out/r3_final_tests_$i.log)"
done
bash scripts/gpu_r3_soak.sh > /dev/null 2>&1; grep -c "SOAK OK" gpurun_out/r3_soak.txt; grep -i "fault\|FAILED" gpurun_out/r3_soak.txt | head -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r3_bench_default.json"))
print("config 2", j["value"], j["ms_per_step"], j["roofline"]["frac"])
for k, v in j["other_configs"].items():
    print(k, v.get("value"), v.get("ms_per_step"), (v.get("config") or {}).get("kernel"), v.get("max_abs_err_vs_oracle"), v.get("error"))
PY
