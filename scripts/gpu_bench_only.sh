#!/bin/bash
# bench-only GPU check (no pytest): the headline shapes for one kernel; parity of stream 0 is still checked by bench.py
cd "$GRAFT_REPO_ROOT"
K=${1:-a1_mfma}
for args in "--launch block --steps 2000 --warmup 200" "--launch resident --steps 2000 --warmup 200" "--launch resident --streams 1024 --steps 500 --warmup 50" "--launch resident --streams 4096 --steps 300 --warmup 30"; do
  python bench.py --kernel $K $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['config']['kernel'], j['config']['launch'], 'streams', j['config']['streams_per_gpu'], 'xRT', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'frac', j['roofline']['frac'], 'err', j['max_abs_err_vs_oracle'])
"
done
