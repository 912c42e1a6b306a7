#!/bin/bash
# nam_wn_reg_kernel with LDS-resident rings + width groups in one launch: the tests that touch it first, then the whole
# suite, then configs 5 and 4 (default and driver-shaped; config 5 also on the VALU kernel for the A/B).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-wrring}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lds_ring or register_resident or slimmable or wavenet_matches_oracle or persistent_block_mode_matches" > gpurun_out/pytest_wr_$TAG.log 2>&1
echo "wr tests rc=$?"; tail -15 gpurun_out/pytest_wr_$TAG.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_all_$TAG.log 2>&1; echo "all rc=$?"; tail -8 gpurun_out/pytest_all_$TAG.log
for c in 5 4; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 > gpurun_out/bench_c${c}_driver_$TAG.json 2> gpurun_out/bench_c${c}_driver_$TAG.err
  timeout 900 python bench.py --config $c > gpurun_out/bench_c${c}_$TAG.json 2> gpurun_out/bench_c${c}_$TAG.err
  timeout 900 python bench.py --config $c --persistent 0 --no-side-runs --no-cpu-baseline > gpurun_out/bench_c${c}_np_$TAG.json 2> gpurun_out/bench_c${c}_np_$TAG.err
  [ $c = 5 ] && timeout 900 python bench.py --config $c --kernel a1 --no-side-runs --no-cpu-baseline > gpurun_out/bench_c${c}_a1_$TAG.json 2> gpurun_out/bench_c${c}_a1_$TAG.err
  python - <<PY
import json
for f in ("gpurun_out/bench_c${c}_driver_$TAG.json", "gpurun_out/bench_c${c}_$TAG.json", "gpurun_out/bench_c${c}_np_$TAG.json", "gpurun_out/bench_c${c}_a1_$TAG.json"):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("config $c", f.split("/")[-1], "value", j["value"], "us/step", round(j["ms_per_step"] * 1e3, 2), "kernel", j["config"]["kernel"], "persist", j["config"].get("persistent_block_mode"),
              "frac", j["roofline"]["frac"], "resident", (j.get("resident_launch") or {}).get("value"), "lat_p50", (j.get("latency_us") or {}).get("p50"),
              "err", j["max_abs_err_vs_oracle"])
    except Exception as e:
        print("config $c", f, "FAILED", e)
        import subprocess; print(open(f.replace(".json", ".err")).read()[-1500:])
PY
done
