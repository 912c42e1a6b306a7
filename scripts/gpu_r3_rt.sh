#!/bin/bash
# the adapter's per-buffer round trips (persistent session / launch per buffer), all three models
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
{
for m in wavenet_a1_standard lstm wavenet_a2_max A2; do
  for buf in 64 128 256; do
    for np_ in 0 1; do
      echo "== benchmodel $m buffer $buf NAM_HIP_NO_PERSISTENT=$np_ (1 stream, nam::DSP::process)"
      NAM_HIP_NO_PERSISTENT=$np_ timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --buffer $buf 2>&1 | grep -i "round trip\|x real\|ms$" | head -3
    done
  done
  for buf in 64 256; do
    echo "== benchmodel $m 256 streams host buffers, buffer $buf, persistent / launch per buffer"
    timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams 256 --buffer $buf 2>&1 | grep -i "round trip\|x real" | head -3
    NAM_HIP_NO_PERSISTENT=1 timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams 256 --buffer $buf 2>&1 | grep -i "round trip\|x real" | head -3
  done
done
} 2>&1 | tee gpurun_out/r3_adapter_roundtrip.txt
