#!/bin/bash
# round 3: nam_wn_reg_kernel compiled per model (wr_jit.cpp): the feature-rich fuzz models, the edge-case model, every test
# of the register-resident kernel; then config 4 / 5 benches and a de-fit table (exact AOT vs per-model compile vs
# run-time flags (NAM_HIP_JIT=0) vs interpreter) on models that are NOT wavenet_a2_max
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ls neuralampmodelercore_amd/lib/jit | wc -l
timeout 1500 python -m pytest tests/test_gpu_breadth.py tests/test_gpu_parity.py -k "featured or per_model or register_resident or lds_ring or wavenet_matches or fuzzed or container or post_stack or prewarm_cache or persistent_block_mode_matches" -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/r3_jit_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_jit_tests.log
tail -8 gpurun_out/r3_jit_tests.log
ls neuralampmodelercore_amd/lib/jit | wc -l
timeout 900 python tools/defit_table.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_defit_table.txt
