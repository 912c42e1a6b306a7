#!/bin/bash
# One GPU call that answers "is the tree good": the GPU suite, the driver's bench command (line size, parity, every config), the
# issue-port and synchronize-tail probes, the official sizes. Writes gpurun_out/check_*.{txt,json}.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round_check.sh'
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
( timeout 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 ) > gpurun_out/check_pytest.txt; tail -3 gpurun_out/check_pytest.txt
timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/check_bench_default.out 2> gpurun_out/check_bench_default.err
tail -1 gpurun_out/check_bench_default.out | wc -c; tail -1 gpurun_out/check_bench_default.out
cp gpurun_out/bench_full.json gpurun_out/check_bench_default_full.json
grep PARITY gpurun_out/check_bench_default.err
for p in valu_rate sync_tail; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/$p tools/src/$p.hip 2>/dev/null && /tmp/$p > gpurun_out/check_$p.txt
done
timeout 300 python tools/official_sizes_probe.py 256 2>&1 | grep -v amdgpu.ids > gpurun_out/check_official_sizes_256.txt; cat gpurun_out/check_official_sizes_256.txt
timeout 200 python tools/a1q_timeline.py 20 > gpurun_out/check_a1q_timeline.txt 2>&1
