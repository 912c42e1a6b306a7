#!/bin/bash
# round 4: the tree as shipped — the whole GPU suite, smoke, the driver's bench command (full line), counters of the headline
# kernel's resident launch (settled clocks), its stage timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/r4_final_tests.log 2>&1
echo "suite rc=$? $(tail -1 gpurun_out/r4_final_tests.log)"; grep "^FAILED\|^ERROR" gpurun_out/r4_final_tests.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4_bench_default.json"))
print("config 2", j["value"], j["ms_per_step"], j["config"]["kernel"], {k: j["roofline"].get(k) for k in ("floor_frac", "floor_parts_us", "frac", "traffic", "counter_frac")})
print("  zeros", j.get("zeros_input"), "fast_tanh_off", j.get("fast_tanh_off"), "resident", j.get("resident_launch"))
for k, v in j["other_configs"].items():
    print(k, v.get("value"), v.get("ms_per_step"), (v.get("config") or {}).get("kernel"), v.get("max_abs_err_vs_oracle"), v.get("error"))
print("cpu", j.get("cpu_baseline", {}).get("value"), j.get("cpu_baseline", {}).get("kind"))
print("host_io", {k: (v.get("blocking", {}).get("value"), v.get("tickets", {}).get("value")) for k, v in (j.get("host_io") or {}).items() if isinstance(v, dict)}, (j.get("host_io") or {}).get("error"))
PY
bash scripts/gpu_prof_resident.sh c2_q nam_a1_q_kernel --config 2 2>&1 | tail -45
timeout 300 python tools/a1q_timeline.py 20 400 > gpurun_out/r4_timeline.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_timeline.txt | head -20
