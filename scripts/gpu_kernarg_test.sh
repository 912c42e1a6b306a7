#!/bin/bash
# developer tool: block-mode step time vs host enqueue time per kernel
cd "$GRAFT_REPO_ROOT"
for k in a1_ws a1_mfma; do
  python bench.py --kernel $k --launch block --steps 3000 --warmup 300 --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python -c "
import json
j = json.load(open('/tmp/b.json'))
print('$k', 'block us/step', round(j['ms_per_step']*1e3,2), 'host enqueue us/step', j['host_enqueue_us_per_step'])
"
done
