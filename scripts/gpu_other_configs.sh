#!/bin/bash
# the non-headline BASELINE.json configs (per-GPU shapes), resident and block launches
cd "$GRAFT_REPO_ROOT"
run() {
  python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
c = j['config']
print(c['workload'].split(',')[0], 'streams', c['streams_per_gpu'], c['launch'], c['kernel'], 'xRT', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'err', j['max_abs_err_vs_oracle'])
"
}
run --model lstm --streams 1024 --launch block --steps 500 --warmup 50
run --model lstm --streams 1024 --launch resident --steps 500 --warmup 50
run --model wavenet_a2_max --streams 512 --launch block --steps 300 --warmup 30
run --model wavenet_a2_max --streams 512 --launch resident --steps 300 --warmup 30
run --model slimmable_wavenet --streams 768 --launch block --steps 500 --warmup 50
run --model slimmable_wavenet --streams 768 --launch resident --steps 500 --warmup 50
run --model A2 --streams 256 --launch block --steps 300 --warmup 30
run --model A2 --streams 256 --launch resident --steps 300 --warmup 30
run --model wavenet_a1_standard --streams 256 --kernel a1 --launch resident --steps 1000 --warmup 100
run --model wavenet_a1_standard --streams 256 --kernel generic --launch resident --steps 300 --warmup 30
