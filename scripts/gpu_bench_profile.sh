#!/bin/bash
# Runs on the GPU box (via gpurun): headline bench + rocprofv3 kernel trace + HBM-traffic PMC passes.
# Outputs land in gpurun_out/ (merged back); the summaries worth keeping are copied into profiles/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r1}
KERNEL=${2:-auto}
mkdir -p gpurun_out
python bench.py --kernel $KERNEL --launch block --steps 3000 --warmup 300 > gpurun_out/bench_block_$TAG.json 2> gpurun_out/bench_block_$TAG.err
python bench.py --kernel $KERNEL --launch resident --steps 3000 --warmup 300 --no-cpu-baseline > gpurun_out/bench_resident_$TAG.json 2>/dev/null
cat gpurun_out/bench_block_$TAG.json gpurun_out/bench_resident_$TAG.json
BENCH="python bench.py --kernel $KERNEL --launch block --steps 500 --warmup 50 --no-cpu-baseline --check 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o trace -- $BENCH > gpurun_out/prof_bench_$TAG.json 2> gpurun_out/prof_$TAG.err
# HBM traffic: separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_$TAG -o pmc_fetch -- $BENCH > /dev/null 2> gpurun_out/pmc_fetch_$TAG.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_$TAG -o pmc_write -- $BENCH > /dev/null 2> gpurun_out/pmc_write_$TAG.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/prof_$TAG -o pmc_sq -- $BENCH > /dev/null 2> gpurun_out/pmc_sq_$TAG.err
find gpurun_out/prof_$TAG -type f | head -30
python scripts/summarize_prof.py gpurun_out/prof_$TAG > gpurun_out/prof_summary_$TAG.txt 2>&1; cat gpurun_out/prof_summary_$TAG.txt
