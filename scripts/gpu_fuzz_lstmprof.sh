#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python tools/fuzz_models.py 36 101 > gpurun_out/fuzz_36_101.txt 2>&1; tail -40 gpurun_out/fuzz_36_101.txt | cut -c1-220
D=gpurun_out/prof_lstm_wide
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python bench.py --config 3 --model synth_lstm_h18x2 --persistent 0 --steps 300 --warmup 30 --reps 1 --no-cpu-baseline --no-side-runs --check 0 --spinup-ms 0 > gpurun_out/prof_lstm_wide_bench.json 2> gpurun_out/prof_lstm_wide.err
find $D -name "trace_kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_lstm_wide_h18x2.csv \;
head -3 gpurun_out/kernel_stats_lstm_wide_h18x2.csv | cut -c1-250
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $D -o pmc_inst -- python bench.py --config 3 --model synth_lstm_h18x2 --persistent 0 --steps 300 --warmup 30 --reps 1 --no-cpu-baseline --no-side-runs --check 0 --spinup-ms 0 > /dev/null 2> gpurun_out/prof_lstm_wide_pmc.err
python scripts/summarize_prof.py $D > gpurun_out/prof_summary_lstm_wide_h18x2.txt 2>&1; grep -i "lstm" gpurun_out/prof_summary_lstm_wide_h18x2.txt | cut -c1-400
