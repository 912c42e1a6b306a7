#!/bin/bash
# counters of nam_wn_reg_kernel as the sessions run it (a resident launch of 300 steps: config 4 = two wavefronts per stream, config 5 = one)
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/gpu_profile_config.sh 4 c4_wn_reg2_resident --launch resident > gpurun_out/prof_c4_res.log 2>&1; tail -4 gpurun_out/prof_c4_res.log | cut -c1-260
bash scripts/gpu_profile_config.sh 5 c5_wn_reg_resident --launch resident > gpurun_out/prof_c5_res.log 2>&1; tail -4 gpurun_out/prof_c5_res.log | cut -c1-260
python scripts/resident_counters.py gpurun_out/prof_c4_wn_reg2_resident nam_wn_reg gpurun_out/counters_c4_wn_reg2_resident.json | tail -2 | cut -c1-400
python scripts/resident_counters.py gpurun_out/prof_c5_wn_reg_resident nam_wn_reg gpurun_out/counters_c5_wn_reg_resident.json | tail -2 | cut -c1-400
