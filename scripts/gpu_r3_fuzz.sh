#!/bin/bash
# round 3: 160 seeded random models (multi-array / K-tap / narrow / feature-rich) through every kernel that takes them:
# 64-frame launches, one multi-block launch (the pipelined forms), a persistent session — against the oracle
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 1500 python tools/fuzz_models.py 160 12 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_fuzz_160.txt
tail -5 gpurun_out/r3_fuzz_160.txt
grep -c "^ok" gpurun_out/r3_fuzz_160.txt; grep "^FAIL" gpurun_out/r3_fuzz_160.txt | head -5 | cut -c1-400
