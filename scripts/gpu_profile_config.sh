#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + PMC passes (separate runs, kernel-trace only) of `bench.py --config C`,
# condensed into gpurun_out/prof_summary_<TAG>.txt and a profiles/traffic.json entry printed as JSON.
# Usage: bash scripts/gpu_profile_config.sh CONFIG TAG [extra bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
C=${1:-2}; TAG=${2:-c2}; shift; shift
BENCH="python bench.py --config $C --launch block --steps 300 --warmup 30 --reps 1 --no-cpu-baseline --no-side-runs --check 0 --spinup-ms 0 $@"
D=gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- $BENCH > gpurun_out/prof_bench_$TAG.json 2> gpurun_out/prof_$TAG.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D -o pmc_fetch -- $BENCH > /dev/null 2> gpurun_out/pmc_fetch_$TAG.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D -o pmc_write -- $BENCH > /dev/null 2> gpurun_out/pmc_write_$TAG.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $D -o pmc_sq -- $BENCH > /dev/null 2> gpurun_out/pmc_sq_$TAG.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $D -o pmc_inst -- $BENCH > /dev/null 2> gpurun_out/pmc_inst_$TAG.err
python scripts/summarize_prof.py $D > gpurun_out/prof_summary_$TAG.txt 2>&1
cat gpurun_out/prof_summary_$TAG.txt | cut -c1-400
cp $D/*/trace_kernel_stats.csv gpurun_out/kernel_stats_$TAG.csv 2>/dev/null || find $D -name "trace_kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_$TAG.csv \;
