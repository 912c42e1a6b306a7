#!/bin/bash
# headline model across stream counts, one launch per block and resident
cd "$GRAFT_REPO_ROOT"
for n in 64 128 256 512 1024 2048 4096 8192; do
  steps=$((600000 / n)); [ $steps -gt 2000 ] && steps=2000; [ $steps -lt 60 ] && steps=60
  python bench.py --streams $n --launch block --steps $steps --warmup $((steps / 10)) --no-cpu-baseline --check 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j['resident_launch']
print('streams', $n, 'block: xRT', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'frac', j['roofline']['frac'], '| resident: xRT', r['value'], 'us/step', round(r['ms_per_step']*1e3,2))
"
done
