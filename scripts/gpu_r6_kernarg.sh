cd "$GRAFT_REPO_ROOT"
cp variants/libnam_hip_stag24.so neuralampmodelercore_amd/lib/libnam_hip.so
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v python tools/a1q_timeline.py 20 2>/dev/null | tail -17 | head -5
  HIP_FORCE_DEV_KERNARG=$v python3 bench.py --gpus 1 --no-other-configs --no-side-runs --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', j['config']['kernel'], round(j['ms_per_step']*1e3,3), 'us/step', j['value'], j.get('region_us'), 'err', j['max_abs_err_vs_oracle'])"
done
