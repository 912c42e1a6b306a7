#!/bin/bash
# A2-Lite (A2.nam at ratio 0.2: 3 channels) and other narrow models under AUTO across stream counts (C++ tool, device-resident
# buffers): nam_wn_reg_kernel sessions, their workgroups taking turns beyond the chip's capacity, the VALU kernel beyond that
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_breadth.py tests/test_gpu_parity.py tests/test_container.py -m gpu -q --timeout=600 -p no:cacheprovider -k "bench_shapes or container or a2 or A2 or slimmable or persistent" 2>&1 | tail -2
{
for st in 256 512 1024 2048; do
  echo "== A2-Lite $st streams, AUTO"
  timeout 120 cpp/tools/benchmodel tests/golden/models/A2.nam --slim 0.2 --streams $st --resident 2>&1 | grep "x real\|kernel:\|rror"
done
for m in synth_a1_nano slimmable_wavenet wavenet wavenet_a2_max; do
  for st in 768 768 2048; do
    echo "== $m $st streams, AUTO"
    timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams $st --resident 2>&1 | grep "x real\|kernel:\|rror"
  done
done
echo "== slimmable_wavenet 1500 streams soak (turns)"; timeout 300 python tools/persist_soak.py 1500 300 3 slimmable_wavenet 2>&1 | grep -v amdgpu.ids | tail -2
} 2>&1 | tee gpurun_out/r3_a2lite.txt
