#!/bin/bash
# A2-Lite (A2.nam at ratio 0.2: 3 channels) under AUTO across stream counts (C++ tool, device-resident buffers), after the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -2
{
for st in 256 512 1024 2048; do
  echo "== A2-Lite $st streams, AUTO"
  timeout 120 cpp/tools/benchmodel tests/golden/models/A2.nam --slim 0.2 --streams $st --resident 2>&1 | grep "x real\|kernel:"
done
for m in synth_a1_nano slimmable_wavenet wavenet; do
  for st in 768 2048; do
    echo "== $m $st streams, AUTO"
    timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams $st --resident 2>&1 | grep "x real\|kernel:"
  done
done
} 2>&1 | tee gpurun_out/r3_a2lite.txt
