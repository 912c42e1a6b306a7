#!/bin/bash
# Ticketed host buffers (nam_hip_batch_submit_f32 / nam_hip_batch_wait_f32): the tests, then the C++ adapter's feeder loop
# (cpp/tools/benchmodel --streams 256 --in-flight 4) beside the blocking loop, then bench.py's host_io block
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 900 python -m pytest tests/test_gpu_tickets.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r4_tickets_tests.log
{
for m in wavenet_a1_standard A2; do
  for buf in 64 256 1024 4096; do
    echo "== benchmodel $m 256 streams host buffers, buffer $buf: blocking"
    timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams 256 --buffer $buf 2>&1 | grep -i "x real" | head -2
    for d in 4 8 16; do
      echo "== benchmodel $m 256 streams host buffers, buffer $buf: $d in flight"
      timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams 256 --buffer $buf --in-flight $d 2>&1 | grep -i "x real\|p50\|Error" | head -3
    done
  done
done
} 2>&1 | tee gpurun_out/adapter_tickets.txt
timeout 600 python - <<'PY' 2>&1 | tee gpurun_out/r4_host_io.json
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
print(json.dumps(bench.host_io(os.path.join("tests", "golden", "models", "wavenet_a1_standard.nam"), 256, True)))
PY
