#!/bin/bash
# round 3, first contact: the new parity-breadth tests (feature-rich models, every-stream persistent bench shapes,
# adapter / multi-device / LSTM scratch / sequence rebase) and the extended fuzz tool
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_breadth.py -m gpu -q -x --timeout=600 -p no:cacheprovider > gpurun_out/r3_breadth.log 2>&1
echo "breadth rc=$?" >> gpurun_out/r3_breadth.log
tail -30 gpurun_out/r3_breadth.log
timeout 600 python tools/fuzz_models.py 24 31 > gpurun_out/r3_fuzz.log 2>&1
tail -8 gpurun_out/r3_fuzz.log
