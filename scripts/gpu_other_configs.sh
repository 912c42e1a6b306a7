#!/bin/bash
# the non-headline BASELINE.json configs (per-GPU shapes) and the A2 container, block and resident launches
cd "$GRAFT_REPO_ROOT"
run() {
  python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
c = j['config']
print(c['workload'].split(',')[0], 'streams', c['streams_per_gpu'], c['launch'], c['kernel'], 'xRT', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'err', j['max_abs_err_vs_oracle'])
"
}
for l in block resident; do
  run --model lstm --streams 1024 --launch $l --steps 500 --warmup 50
  run --model wavenet_a2_max --streams 512 --launch $l --steps 300 --warmup 30
  run --model slimmable_wavenet --streams 768 --launch $l --steps 500 --warmup 50
  run --model A2 --streams 256 --launch $l --steps 300 --warmup 30
  run --model A2 --streams 2048 --launch $l --steps 100 --warmup 10
done
run --model wavenet_a1_standard --streams 256 --kernel a1 --launch block --steps 500 --warmup 50
run --model wavenet_a1_standard --streams 4096 --kernel a1 --launch resident --steps 100 --warmup 10
