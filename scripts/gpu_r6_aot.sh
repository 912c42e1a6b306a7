#!/bin/bash
# round 6: nam_wn_reg_kernel WITHOUT the per-model compile (NAM_HIP_JIT=0: the ahead-of-time kernels walking the program, run-time-flag
# shapes) and with the built-in kernels first (NAM_HIP_WR_PROGRAM=0): the featured models, the bench shapes and a fuzz run against the oracle
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export NAM_HIP_PERSIST_TIMEOUT_MS=8000
for env in "NAM_HIP_JIT=0" "NAM_HIP_WR_PROGRAM=0"; do
  echo "== $env"
  env $env timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_breadth.py -k "featured or (bench_shapes and (4 or 5 or 24 or 25))" 2>&1 | tail -4
  env $env timeout 600 python tools/fuzz_models.py 48 4242 2>&1 | grep -v "^ok" | tail -6
done
