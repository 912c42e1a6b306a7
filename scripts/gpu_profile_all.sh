#!/bin/bash
# Round profile set: rocprofv3 kernel trace + PMC passes (scripts/gpu_profile_config.sh) for every BASELINE config with
# one launch per step (a dispatch = a step: per-dispatch counters are per-step counters), plus the op interpreter and
# the VALU kernel. Summaries land in gpurun_out/prof_summary_<tag>.txt and gpurun_out/kernel_stats_<tag>.csv.
cd "$GRAFT_REPO_ROOT"
for spec in "2 c2_p2 --persistent 0 --kernel a1_il" "3 c3_lstm_row --persistent 0" "4 c4_wn_reg --persistent 0" "5 c5_slimmable --persistent 0" "4 c4_generic --persistent 0 --kernel generic --steps 100 --warmup 10" "2 c2_valu --persistent 0 --kernel a1 --steps 100 --warmup 10"; do
  set -- $spec
  echo "=== $spec"
  timeout 600 bash scripts/gpu_profile_config.sh "$@" > gpurun_out/prof_run_$2.log 2>&1
  tail -3 gpurun_out/prof_run_$2.log | cut -c1-200
done
