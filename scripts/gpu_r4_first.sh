#!/bin/bash
# round 4, first contact of nam_a1_q_kernel with the hardware: the block-by-block probe against nam_a1_p2_kernel, its timing
# against nam_a1_p4_kernel, the A1 part of the GPU suite, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 300 python tools/a1q_probe.py 5 > gpurun_out/r4_probe.txt 2>&1; echo "probe rc=$?"; tail -25 gpurun_out/r4_probe.txt
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider -x -k "a1 or A1 or bench_shapes or pipelined or persistent or il_kernel or smoke" > gpurun_out/r4_first_tests.log 2>&1
echo "a1 tests rc=$? $(tail -3 gpurun_out/r4_first_tests.log)"
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_first.json 2> gpurun_out/r4_bench_first.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r4_bench_first.json"))
    print("config 2", j["value"], j["ms_per_step"], j["config"], j["roofline"])
except Exception as e:
    print("bench json:", e)
PY
