#!/bin/bash
# round 4: the q kernel's probe (correctness + steady timing), the A1 part of the suite + the soak tests, driver-shaped bench q vs p4, the stage timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 300 python tools/a1q_probe.py 5 > gpurun_out/r4_probe.txt 2>&1; echo "probe rc=$?"; grep -v "per block" gpurun_out/r4_probe.txt | tail -6; grep "max |q" gpurun_out/r4_probe.txt | cut -c1-100
timeout 1200 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider -k "a1 or A1 or bench_shapes or pipelined or persistent or il_kernel or smoke or reference or long_session" > gpurun_out/r4_check_tests.log 2>&1
echo "tests rc=$? $(tail -3 gpurun_out/r4_check_tests.log)"; grep "^FAILED\|^ERROR" gpurun_out/r4_check_tests.log | head
for q in 1 0; do
  NAM_HIP_A1Q=$q timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-side-runs --no-cpu-baseline > gpurun_out/r4_bench_q$q.json 2> gpurun_out/r4_bench_q$q.err; echo "bench q=$q rc=$?"
  python - gpurun_out/r4_bench_q$q.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("  driver shape:", j["config"]["kernel"], j["value"], "xRT", j["ms_per_step"] * 1e3, "us/step;", j.get("region_us"), "parity", j.get("max_abs_err_vs_oracle"))
PY
done
timeout 300 python tools/a1q_timeline.py 20 400 > gpurun_out/r4_timeline.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4_timeline.txt
