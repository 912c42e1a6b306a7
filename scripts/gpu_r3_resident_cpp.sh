#!/bin/bash
# the C++ adapter with device-resident buffers (BatchDSP::process_device + flush), next to its blocking host-buffer form
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
{
for m in wavenet_a1_standard A2 wavenet_a2_max lstm; do
  for st in 256 512; do
    echo "== $m, $st streams"
    timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams $st --resident 2>&1 | grep "x real"
    timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams $st 2>&1 | grep "x real"
  done
done
} 2>&1 | tee gpurun_out/r3_cpp_resident.txt
