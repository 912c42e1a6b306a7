bash scripts/gpu_r5_variants.sh > gpurun_out/r5_variants.txt 2>&1; cat gpurun_out/r5_variants.txt
cd "$GRAFT_REPO_ROOT"
python tools/a1q_timeline.py 20 > gpurun_out/r5_a1q_timeline.txt 2>&1; tail -18 gpurun_out/r5_a1q_timeline.txt
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r5_pytest2.txt; tail -3 gpurun_out/r5_pytest2.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/src/valu_rate.hip 2>/dev/null && /tmp/valu_rate > gpurun_out/r5_valu_rate.txt; tail -11 gpurun_out/r5_valu_rate.txt
