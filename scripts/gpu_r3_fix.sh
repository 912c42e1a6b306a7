#!/bin/bash
# round 3: the whole GPU suite, the adapter's per-buffer round trips (persistent session / launch per buffer), the default
# bench line; with PROFILE=1 also the counters of the pipelined kernel (resident launch: one dispatch = K steps) and of
# config 4, and kernel traces of the default command of configs 2-5
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NAM_HIP_PERSIST_TIMEOUT_MS=8000
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/r3_fix_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_fix_tests.log
tail -12 gpurun_out/r3_fix_tests.log
{
for m in wavenet_a1_standard lstm wavenet_a2_max; do
  for buf in 64 128 256; do
    for np_ in 0 1; do
      echo "== benchmodel $m buffer $buf NAM_HIP_NO_PERSISTENT=$np_ (1 stream, nam::DSP::process)"
      NAM_HIP_NO_PERSISTENT=$np_ timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --buffer $buf 2>&1 | grep -i "round trip\|x real\|ms$" | head -3
    done
  done
  for buf in 64 256; do
    echo "== benchmodel $m 256 streams host buffers, buffer $buf, persistent / launch per buffer"
    timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams 256 --buffer $buf 2>&1 | grep -i "round trip\|x real" | head -3
    NAM_HIP_NO_PERSISTENT=1 timeout 120 cpp/tools/benchmodel tests/golden/models/$m.nam --streams 256 --buffer $buf 2>&1 | grep -i "round trip\|x real" | head -3
  done
done
} 2>&1 | tee gpurun_out/r3_adapter_roundtrip.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
cut -c1-600 gpurun_out/r3_bench_default.json
[ "${PROFILE:-0}" = 1 ] || exit 0
bash scripts/gpu_profile_config.sh 2 c2_p4_resident --launch resident > gpurun_out/prof_c2_p4.log 2>&1; tail -8 gpurun_out/prof_c2_p4.log | cut -c1-300
bash scripts/gpu_profile_config.sh 4 c4_wn_reg_jit --persistent 0 > gpurun_out/prof_c4_jit.log 2>&1; tail -8 gpurun_out/prof_c4_jit.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in 2 3 4 5; do
  D=gpurun_out/ptrace3_c${c}
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o trace -- python bench.py --config $c --no-side-runs --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 > gpurun_out/ptrace3_bench_c${c}.json 2> gpurun_out/ptrace3_c${c}.err
  find $D -name "trace_kernel_stats.csv" -exec cp {} gpurun_out/r3_persistent_kernel_stats_c${c}.csv \;
  head -3 gpurun_out/r3_persistent_kernel_stats_c${c}.csv | cut -c1-200
done
