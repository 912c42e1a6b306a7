#!/bin/bash
# nam_kq_kernel as the A2 pipeline kernel: the 500-stream soak (workgroups catch up with the host, leave and restart), then the
# whole GPU suite — with nam_kq_kernel if the soak is clean, with NAM_HIP_KQ=0 (nam_kp_kernel) otherwise
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 120 python tools/persist_soak.py 500 1000 6 A2 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee gpurun_out/r3_kq_soak500.txt
timeout 60 python tools/kq_probe.py 2>&1 | grep "prewarm=" | tee -a gpurun_out/r3_kq_soak500.txt
if grep -q "SOAK OK" gpurun_out/r3_kq_soak500.txt; then echo "suite with nam_kq_kernel"; else export NAM_HIP_KQ=0; echo "suite with NAM_HIP_KQ=0"; fi
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/r3_kq_final_tests.log 2>&1
echo "suite rc=$? NAM_HIP_KQ=${NAM_HIP_KQ:-unset} $(tail -1 gpurun_out/r3_kq_final_tests.log)" | tee -a gpurun_out/r3_kq_soak500.txt
