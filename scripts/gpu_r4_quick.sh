cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 200 python tools/a1q_time.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/a1q_probe.py 5 2>&1 | grep "PROBE\|resident launch" 
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-side-runs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('   driver shape', j['config']['kernel'], round(j['ms_per_step']*1e3,3), 'us/step', j['region_us'], 'err', j['max_abs_err_vs_oracle'], j['roofline'].get('floor_frac'), j['roofline'].get('floor_parts_us'))
"
