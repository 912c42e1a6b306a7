cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/gpu_prof_resident.sh c2_q nam_a1_q_kernel --config 2 2>&1 | tail -45
# the driver's own command under the kernel trace: the session launches' durations next to the line's ms_per_step
D=gpurun_out/prof_driver
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-side-runs --no-cpu-baseline > gpurun_out/prof_driver_line.json 2> gpurun_out/prof_driver.err
find $D -name "trace_kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_driver.csv \;
head -5 gpurun_out/kernel_stats_driver.csv; tail -1 gpurun_out/prof_driver_line.json | cut -c1-400
python - <<'PY'
import csv, glob, statistics
# the SESSION instantiation (PERSIST = true) only: the plain one in the same trace is the bench's untimed spin-up on its scratch batch
rows=[r for r in csv.DictReader(open(glob.glob("gpurun_out/prof_driver/**/trace_kernel_trace.csv", recursive=True)[0])) if "nam_a1_q_kernel<3, false, true" in r["Kernel_Name"]]
d=sorted(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
short=[x for x in d if x < 90000]; long_=[x for x in d if x >= 90000]
print("nam_a1_q_kernel session dispatches", len(d), "| 5-buffer warm-up launches: n", len(short), "median", statistics.median(short)/1e3 if short else None, "us | 20-buffer launches: n", len(long_), "median", statistics.median(long_)/1e3 if long_ else None, "min", long_[0]/1e3 if long_ else None, "max", long_[-1]/1e3 if long_ else None, "us =", (statistics.median(long_)/20e3 if long_ else None), "us per buffer in-kernel")
PY
