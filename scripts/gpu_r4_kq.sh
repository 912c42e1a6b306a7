#!/bin/bash
# round 4: nam_kq_kernel: its tests + soaks, the A2 bench line (driver shape and 500-step regions), counters of a resident launch
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 900 python -m pytest tests -m gpu -q -x --timeout=300 -p no:cacheprovider -k "A2 or a2 or container or kq or ktap or k_tap or long_session or bench_shapes" > gpurun_out/r4_kq_tests.log 2>&1
echo "a2 tests rc=$? $(tail -1 gpurun_out/r4_kq_tests.log)"; grep "^FAILED\|^ERROR" gpurun_out/r4_kq_tests.log | head
for shape in "--steps 20 --warmup 5" "--steps 500 --warmup 50 --brief"; do
timeout 600 python3 bench.py --model A2 --streams 256 --gpus 1 $shape --no-other-configs --no-side-runs --no-cpu-baseline 2> gpurun_out/r4_bench_a2.err | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('A2', j['steps'], 'steps:', j['value'], 'xRT', round(j['ms_per_step']*1e3, 3), 'us/step', j['config']['kernel'], 'err', j['max_abs_err_vs_oracle'])
"
done
[ "$1" = "prof" ] && bash scripts/gpu_prof_resident.sh a2_kq nam_kq_kernel --model A2 --streams 256 2>&1 | grep -v "SQC_\|IFETCH\|BRANCH\|SMEM\|MISC\|SCA\|BUSY_C\|LEVEL" | tail -24
