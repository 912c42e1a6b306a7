#!/bin/bash
# Runs on the GPU box (via gpurun): headline bench in both launch modes + rocprofv3 kernel trace.
# Outputs land in gpurun_out/ (merged back); summaries to keep are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r1}
mkdir -p gpurun_out
python bench.py --launch block --steps 2000 --warmup 200 > gpurun_out/bench_block_$TAG.json 2> gpurun_out/bench_block_$TAG.err
python bench.py --launch resident --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/bench_resident_$TAG.json 2>/dev/null
python bench.py --launch resident --streams 4096 --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_resident4096_$TAG.json 2>/dev/null
cat gpurun_out/bench_*_$TAG.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o a1 -- python bench.py --launch block --steps 500 --warmup 50 --no-cpu-baseline --check 0 > gpurun_out/prof_bench_$TAG.json 2> gpurun_out/prof_$TAG.err
find gpurun_out/prof_$TAG -type f | head -20
