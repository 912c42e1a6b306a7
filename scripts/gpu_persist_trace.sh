#!/bin/bash
# Timeline of a short persistent-mode region: kernel dispatch start/end (kernel-trace) against the HIP API calls
# (hip-runtime-trace) on one clock. Output: gpurun_out/ptrace/*.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ptrace
timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d gpurun_out/ptrace -o pt -- \
  python bench.py --persistent 1 --steps 20 --warmup 5 --reps 3 --no-side-runs --no-cpu-baseline --spinup-ms 0 > gpurun_out/ptrace/bench.json 2> gpurun_out/ptrace/err.log
echo rc=$?
ls -la gpurun_out/ptrace | head; find gpurun_out/ptrace -name "*.csv" | head
python - <<'PY'
import csv, glob
kt = [f for f in glob.glob('gpurun_out/ptrace/**/*kernel_trace.csv', recursive=True)]
ht = [f for f in glob.glob('gpurun_out/ptrace/**/*hip_api_trace.csv', recursive=True)]
print(kt, ht)
rows = []
for f in kt:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), 'K', r['Kernel_Name'][:40], int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
for f in ht:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), 'H', r['Function'], int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
rows.sort()
# the last 140 events before the end of the timed regions: find p2 kernels
idx = [i for i, r in enumerate(rows) if r[1] == 'K' and 'p2' in r[2]]
print('p2 launches', len(idx))
if idx:
    lo = max(0, idx[-1] - 160)
    t0 = rows[lo][0]
    for r in rows[lo: idx[-1] + 12]:
        print(f"{(r[0]-t0)/1e3:10.1f} us {r[1]} {r[2]:42s} {r[3]/1e3:8.1f}")
PY
