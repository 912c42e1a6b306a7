#!/bin/bash
# round 5, first GPU call: the suite, the driver's command (line size!), host-wait A/B for the driver shape, the issue microbench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r5_pytest.txt
tail -3 gpurun_out/r5_pytest.txt
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_default.out 2> gpurun_out/r5_bench_default.err
tail -1 gpurun_out/r5_bench_default.out | wc -c
tail -1 gpurun_out/r5_bench_default.out
cp gpurun_out/bench_full.json gpurun_out/r5_bench_default_full.json
Q() { python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', '$1', j['config']['kernel'], round(j['ms_per_step']*1e3,3), 'us/step', j['value'], j.get('region_us'), 'err', j['max_abs_err_vs_oracle'])
"; }
B="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-side-runs --no-cpu-baseline"
for i in 1 2; do
timeout 200 $B 2>/dev/null | Q base
HSA_ENABLE_INTERRUPT=0 timeout 200 $B 2>/dev/null | Q hsa_interrupt_0
done
ROC_ACTIVE_WAIT_TIMEOUT=1000 timeout 200 $B 2>/dev/null | Q roc_active_wait_1000
GPU_MAX_HW_QUEUES=8 timeout 200 $B 2>/dev/null | Q max_hw_queues_8
HIP_LAUNCH_BLOCKING=0 AMD_SERIALIZE_KERNEL=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 200 $B 2>/dev/null | Q plain_again
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/src/valu_rate.hip 2>/dev/null && /tmp/valu_rate > gpurun_out/r5_valu_rate.txt; tail -16 gpurun_out/r5_valu_rate.txt
