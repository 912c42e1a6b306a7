#!/bin/bash
# The whole -m gpu suite plus the default bench line of the configs given (default: 4 5).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-q}; shift
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_all_$TAG.log 2>&1; echo "all rc=$?"; tail -12 gpurun_out/pytest_all_$TAG.log
for c in ${@:-4 5}; do
  timeout 900 python bench.py --config $c > gpurun_out/bench_c${c}_$TAG.json 2> gpurun_out/bench_c${c}_$TAG.err
  python - <<PY
import json
f = "gpurun_out/bench_c${c}_$TAG.json"
try:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print("config $c", "value", j["value"], "us/step", round(j["ms_per_step"] * 1e3, 2), "kernel", j["config"]["kernel"], "persist", j["config"].get("persistent_block_mode"),
          "frac", j["roofline"]["frac"], "resident", (j.get("resident_launch") or {}).get("value"), "err", j["max_abs_err_vs_oracle"])
except Exception as e:
    print("config $c FAILED", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
done
