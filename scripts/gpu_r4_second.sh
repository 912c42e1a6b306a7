#!/bin/bash
# round 4: vector-issue microbenchmark, the q kernel after the prologue / poll changes (probe incl. timing), driver-shaped bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 120 tools/src/valu_rate > gpurun_out/r4_valu_rate.txt 2>&1; cat gpurun_out/r4_valu_rate.txt
timeout 300 python tools/a1q_probe.py 5 > gpurun_out/r4_probe2.txt 2>&1; echo "probe rc=$?"; grep -v "per block" gpurun_out/r4_probe2.txt | tail -8
for q in 1 0; do
  NAM_HIP_A1Q=$q timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-side-runs --no-cpu-baseline > gpurun_out/r4_bench2_q$q.json 2> gpurun_out/r4_bench2_q$q.err; echo "bench q=$q rc=$?"
  python - gpurun_out/r4_bench2_q$q.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print("  driver shape:", j["config"]["kernel"], j["value"], "xRT", j["ms_per_step"] * 1e3, "us/step; resident:", j.get("resident_launch"))
PY
done
