#!/bin/bash
# round 4: SIMD placement probe, the q kernel with sub-blocks (probe incl. timing), A1 part of the suite, driver-shaped bench q vs p4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 60 tools/src/simd_map > gpurun_out/r4_simd_map.txt 2>&1; head -12 gpurun_out/r4_simd_map.txt
timeout 300 python tools/a1q_probe.py 5 > gpurun_out/r4_probe3.txt 2>&1; echo "probe rc=$?"; grep -v "per block" gpurun_out/r4_probe3.txt | tail -8; grep "max |q" gpurun_out/r4_probe3.txt | cut -c1-90
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider -x -k "a1 or A1 or bench_shapes or pipelined or persistent or il_kernel or smoke or reference" > gpurun_out/r4_third_tests.log 2>&1
echo "a1 tests rc=$? $(tail -3 gpurun_out/r4_third_tests.log)"
for q in 1 0; do
  NAM_HIP_A1Q=$q timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-side-runs --no-cpu-baseline > gpurun_out/r4_bench3_q$q.json 2> gpurun_out/r4_bench3_q$q.err; echo "bench q=$q rc=$?"
  python - gpurun_out/r4_bench3_q$q.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("  driver shape:", j["config"]["kernel"], j["value"], "xRT", j["ms_per_step"] * 1e3, "us/step; resident:", j.get("resident_launch"))
PY
done
