#!/bin/bash
# A2.nam (A2-Full): VALU kernel vs the K-tap MFMA kernel at several stream counts
cd "$GRAFT_REPO_ROOT"
for n in 256 1024 2048 4096; do
  for k in a1 a1_mfma; do
    for l in block resident; do
      python bench.py --model A2 --kernel $k --streams $n --launch $l --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('A2 streams $n $k $l', 'kernel', j['config']['kernel'], 'xRT', j['value'], 'us/step', round(j['ms_per_step']*1e3,2), 'err', j['max_abs_err_vs_oracle'])
"
    done
  done
done
