#!/bin/bash
# Second half of round 2: counters of nam_wn_reg_kernel on configs 5 and 4 (one launch per step, so per-dispatch = per-step)
# and kernel traces of the default (persistent) bench command of configs 4 and 5.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
bash scripts/gpu_profile_config.sh 5 c5_wn_reg --persistent 0 > gpurun_out/prof_c5.log 2>&1; tail -8 gpurun_out/prof_c5.log | cut -c1-300
bash scripts/gpu_profile_config.sh 4 c4_wn_reg --persistent 0 > gpurun_out/prof_c4.log 2>&1; tail -8 gpurun_out/prof_c4.log | cut -c1-300
for c in 4 5; do
  D=gpurun_out/ptrace_c${c}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python bench.py --config $c --no-side-runs --no-cpu-baseline --reps 3 > gpurun_out/ptrace_bench_c${c}.json 2> gpurun_out/ptrace_c${c}.err
  find $D -name "trace_kernel_stats.csv" -exec cp {} gpurun_out/persistent_kernel_stats_c${c}.csv \;
  head -3 gpurun_out/persistent_kernel_stats_c${c}.csv | cut -c1-200
  tail -c 400 gpurun_out/ptrace_bench_c${c}.json | head -c 200; echo
done
