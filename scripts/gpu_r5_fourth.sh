#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/sync_tail tools/src/sync_tail.hip 2>/dev/null && /tmp/sync_tail > gpurun_out/r5_sync_tail.txt; cat gpurun_out/r5_sync_tail.txt
Q() { python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', '$1', j['config']['kernel'], round(j['ms_per_step']*1e3,3), 'us/step', j['value'], j.get('region_us'), 'err', j['max_abs_err_vs_oracle'])
"; }
B="python3 bench.py --gpus 1 --no-other-configs --no-side-runs --no-cpu-baseline"
for i in 1 2; do timeout 200 $B --steps 20 --warmup 5 2>/dev/null | Q driver; done
timeout 200 $B --steps 500 --warmup 50 --brief 2>/dev/null | Q steady
timeout 200 $B --steps 200 --warmup 20 --persistent 0 --kernel a1_mfma 2>/dev/null | Q "lone buffers a1_mfma"
timeout 200 $B --steps 200 --warmup 20 --persistent 0 --kernel a1_il 2>/dev/null | Q "lone buffers a1_il(p2)"
timeout 300 python tools/official_sizes_probe.py 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_official_sizes_256_padded.txt; cat gpurun_out/r5_official_sizes_256_padded.txt
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 > gpurun_out/r5_pytest3.txt; cat gpurun_out/r5_pytest3.txt
