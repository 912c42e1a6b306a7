#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/a1q_timeline.py 20 100 400 > gpurun_out/r4_timeline.txt 2>&1; cat gpurun_out/r4_timeline.txt | grep -v amdgpu.ids
