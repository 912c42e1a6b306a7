#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/sync_tail tools/src/sync_tail.hip 2>/dev/null && /tmp/sync_tail > gpurun_out/r5_sync_tail.txt; cat gpurun_out/r5_sync_tail.txt
timeout 300 python tools/official_sizes_probe.py 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_official_sizes_256.txt; cat gpurun_out/r5_official_sizes_256.txt
timeout 600 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -25 > gpurun_out/r5_pytest_durations.txt; cat gpurun_out/r5_pytest_durations.txt
