cd "$GRAFT_REPO_ROOT"; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
Q() { python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  ', '$1', j['config']['kernel'], round(j['ms_per_step']*1e3,3), 'us/step', j['value'], 'err', j['max_abs_err_vs_oracle'])
"; }
B="python3 bench.py --gpus 1 --no-other-configs --no-side-runs --no-cpu-baseline --brief --steps 500 --warmup 50"
for i in 1 2; do
timeout 200 $B --config 5 2>/dev/null | Q "config 5 (two per SIMD allowed)"
NAM_HIP_MAX_STAGES=1 timeout 200 $B --config 5 2>/dev/null | Q "config 5 MAX_STAGES=1 (one wave per stream)"
done
timeout 200 $B --config 5 --streams 512 2>/dev/null | Q "config 5 at 512 streams"
timeout 200 $B --config 5 --streams 1024 2>/dev/null | Q "config 5 at 1024 streams"
timeout 200 $B --model synth_a1_nano --streams 768 2>/dev/null | Q "nano 768"
timeout 200 $B --model wavenet --streams 768 2>/dev/null | Q "wavenet.nam 768"
timeout 600 python -m pytest tests -q -m gpu -x -k "wn_reg or register_resident or bench_shapes or persistent or soak or slimmable or container" 2>&1 | tail -3
