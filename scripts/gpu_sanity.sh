#!/bin/bash
# Re-entry sanity run: the whole -m gpu suite and the driver-shaped bench line of config 2 on a fresh box.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-sanity}
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_$TAG.log
for c in ${CONFIGS:-2 5}; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 > gpurun_out/bench_c${c}_driver_$TAG.json 2> gpurun_out/bench_c${c}_driver_$TAG.err
  tail -c 1500 gpurun_out/bench_c${c}_driver_$TAG.json | head -c 600; echo
done
