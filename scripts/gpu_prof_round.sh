cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/gpu_prof_resident.sh c2_q nam_a1_q_kernel --config 2 2>&1 | tail -45
# the driver's own command under the kernel trace: the session launches' durations next to the line's ms_per_step
D=gpurun_out/prof_driver
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs --no-side-runs --no-cpu-baseline > gpurun_out/prof_driver_line.json 2> gpurun_out/prof_driver.err
find $D -name "trace_kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_driver.csv \;
head -5 gpurun_out/kernel_stats_driver.csv; tail -1 gpurun_out/prof_driver_line.json | cut -c1-400
python - <<'PY'
import csv, glob
rows=[r for r in csv.DictReader(open(glob.glob("gpurun_out/prof_driver/**/trace_kernel_trace.csv", recursive=True)[0])) if "nam_a1_q_kernel" in r["Kernel_Name"]]
d=sorted(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows)
import statistics
short=[x for x in d if x < 60000]; long_=[x for x in d if x >= 60000]
print("nam_a1_q_kernel dispatches", len(d), "| 5-buffer warm-up launches: median", statistics.median(short)/1e3 if short else None, "us | 20-buffer launches: n", len(long_), "median", statistics.median(long_)/1e3 if long_ else None, "us =", (statistics.median(long_)/20e3 if long_ else None), "us per buffer in-kernel")
PY
