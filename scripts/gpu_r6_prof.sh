#!/bin/bash
# round 6: counters of the resident launches of configs 4, 5 (nam_wn_reg_kernel with the programs compiled in), 2 and A2
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for spec in "c4_wn_reg nam_wn_reg --config 4" "c5_wn_reg nam_wn_reg --config 5" "c2_q nam_a1_q_kernel --config 2" "a2_kq nam_kq_kernel --model A2 --streams 256"; do
  set -- $spec
  bash scripts/gpu_prof_resident.sh "$@" 2>&1 | tail -60 > gpurun_out/prof_summary_$1.txt
  head -3 gpurun_out/prof_summary_$1.txt
done
