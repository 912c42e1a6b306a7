#!/bin/bash
# round 3: long sessions of the pipelined kernels against one launch of their un-pipelined forms (every stream, every frame)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
{
echo "== wavenet_a1_standard 256 streams"; timeout 300 python tools/persist_soak.py 256 3000 3 wavenet_a1_standard 2>&1 | grep -v amdgpu.ids
echo "== wavenet_a1_standard 600 streams (turns)"; timeout 300 python tools/persist_soak.py 600 1000 2 wavenet_a1_standard 2>&1 | grep -v amdgpu.ids
echo "== A2 256 streams"; timeout 300 python tools/persist_soak.py 256 3000 3 A2 2>&1 | grep -v amdgpu.ids
echo "== A2 500 streams (turns)"; timeout 300 python tools/persist_soak.py 500 1000 2 A2 2>&1 | grep -v amdgpu.ids
echo "== wavenet_a2_max 512 streams"; timeout 300 python tools/persist_soak.py 512 3000 3 wavenet_a2_max 2>&1 | grep -v amdgpu.ids
echo "== wavenet_condition_dsp 200 streams"; timeout 300 python tools/persist_soak.py 200 2000 2 wavenet_condition_dsp 2>&1 | grep -v amdgpu.ids
echo "== lstm 1024 streams"; timeout 300 python tools/persist_soak.py 1024 2000 2 lstm 2>&1 | grep -v amdgpu.ids
} 2>&1 | tee gpurun_out/r3_soak.txt
