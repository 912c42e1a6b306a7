#!/bin/bash
# The C++ adapter's per-buffer round trips (host buffer in -> host buffer out through nam::DSP::process / BatchDSP::process_batch):
# session vs launch per buffer, buffers of 64 ... 1,024 frames, one stream and 256 streams
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
{
for m in wavenet_a1_standard A2; do
  for buf in 64 256 1024; do
    for np_ in 0 1; do
      echo "== benchmodel $m buffer $buf NAM_HIP_NO_PERSISTENT=$np_ (1 stream, nam::DSP::process)"
      NAM_HIP_NO_PERSISTENT=$np_ timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --buffer $buf 2>&1 | grep -i "round trip\|x real" | head -3
    done
  done
  for buf in 64 256 1024 4096; do
    echo "== benchmodel $m 256 streams host buffers, buffer $buf, session"
    timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams 256 --buffer $buf 2>&1 | grep -i "round trip\|x real" | head -3
  done
  echo "== benchmodel $m 256 streams, device-resident buffers (process_device + flush)"
  timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams 256 --resident 2>&1 | grep -i "round trip\|x real" | head -3
done
} 2>&1 | tee gpurun_out/adapter_roundtrip.txt
