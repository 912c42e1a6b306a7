#!/bin/bash
# round 4: nam_kq_kernel after the lazy-append change: its tests + soaks, the A2 bench line, counters of a resident launch
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider -k "A2 or a2 or container or kq or ktap or k_tap or long_session or bench_shapes" > gpurun_out/r4_kq_tests.log 2>&1
echo "a2 tests rc=$? $(tail -1 gpurun_out/r4_kq_tests.log)"; grep "^FAILED\|^ERROR" gpurun_out/r4_kq_tests.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python3 bench.py --model A2 --streams 256 --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-side-runs --no-cpu-baseline > gpurun_out/r4_bench_a2.json 2> gpurun_out/r4_bench_a2.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4_bench_a2.json"))
print("A2", j["value"], j["ms_per_step"], j["config"]["kernel"], j["max_abs_err_vs_oracle"], j.get("resident_launch"))
PY
bash scripts/gpu_prof_resident.sh a2_kq nam_kq_kernel --model A2 --streams 256 2>&1 | tail -40
