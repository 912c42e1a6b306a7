#!/bin/bash
# rocprofv3 kernel trace + PMC passes for one bench.py configuration: scripts/gpu_profile_model.sh <tag> <bench.py args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
mkdir -p gpurun_out
python bench.py "$@" --launch block --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/bench_block_$TAG.json 2>/dev/null
cat gpurun_out/bench_block_$TAG.json | cut -c1-300
BENCH="python bench.py $* --launch block --steps 300 --warmup 30 --no-cpu-baseline --check 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o trace -- $BENCH > /dev/null 2> gpurun_out/prof_$TAG.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_$TAG -o pmc_fetch -- $BENCH > /dev/null 2> gpurun_out/pmc_fetch_$TAG.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_$TAG -o pmc_write -- $BENCH > /dev/null 2> gpurun_out/pmc_write_$TAG.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/prof_$TAG -o pmc_sq -- $BENCH > /dev/null 2> gpurun_out/pmc_sq_$TAG.err
python scripts/summarize_prof.py gpurun_out/prof_$TAG > gpurun_out/prof_summary_$TAG.txt 2>&1; cat gpurun_out/prof_summary_$TAG.txt | head -30
