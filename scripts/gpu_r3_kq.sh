#!/bin/bash
# nam_kq_kernel (developer switch NAM_HIP_KQ=1) against the A2 tests, then A2-Full at 256 streams on both pipeline kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
{
NAM_HIP_KQ=1 timeout 400 python -m pytest tests/test_gpu_breadth.py tests/test_gpu_parity.py tests/test_container.py -m gpu -q -p no:cacheprovider -k "a2 or A2 or bench_shapes or container" 2>&1 | grep -v amdgpu.ids | tail -15
for kq in 1 0; do
  echo "== A2-Full 256 streams, NAM_HIP_KQ=$kq"
  NAM_HIP_KQ=$kq timeout 120 cpp/tools/benchmodel tests/golden/models/A2.nam --streams 256 --resident 2>&1 | grep "x real\|kernel:\|rror"
done
} 2>&1 | tee gpurun_out/r3_kq.txt
