#!/bin/bash
# a longer soak: more repetitions (each with random pauses -> the launch leaves and restarts), all pipelined kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
{
for spec in "256 2000 8 wavenet_a1_standard" "700 500 6 wavenet_a1_standard" "256 2000 8 A2" "450 500 6 A2" "512 2000 8 wavenet_a2_max" "256 2000 8 wavenet_a2_max" "100 2000 8 synth_a1_nano" "768 1500 6 slimmable_wavenet" "1024 2000 6 lstm" "37 3000 8 wavenet_condition_dsp"; do
  echo "== $spec"; timeout 400 python tools/persist_soak.py $spec 2>&1 | grep -v amdgpu.ids | grep -v "soak rep" | tail -2
done
} 2>&1 | tee gpurun_out/r3_soak_long.txt
