#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_nano.log 2>&1; echo "all rc=$?"; tail -6 gpurun_out/pytest_nano.log
timeout 600 python tools/official_sizes_probe.py 256 > gpurun_out/official_sizes_256.txt 2>&1; cat gpurun_out/official_sizes_256.txt | tail -12
timeout 600 python tools/official_sizes_probe.py 512 > gpurun_out/official_sizes_512.txt 2>&1; grep nano gpurun_out/official_sizes_512.txt
