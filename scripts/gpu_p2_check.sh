#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-p2}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "interleaved or persistent or headline or golden or smoke or wavenet_matches_oracle or reference_library" > gpurun_out/pytest_p2_$TAG.log 2>&1; echo "p2 tests rc=$?"; tail -5 gpurun_out/pytest_p2_$TAG.log
for extra in "" "--steps 20 --warmup 5"; do
timeout 600 python bench.py --config 2 --no-cpu-baseline --no-side-runs $extra > gpurun_out/bench_c2_$TAG.json 2> gpurun_out/bench_c2_$TAG.err
python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/bench_c2_$TAG.json").read().strip().splitlines()[-1])
    print("config 2 [$extra]", "value", j["value"], "us/step", round(j["ms_per_step"] * 1e3, 2), "kernel", j["config"]["kernel"], "resident", (j.get("resident_launch") or {}).get("value"), "err", j["max_abs_err_vs_oracle"])
except Exception as e:
    print("config 2 FAILED", e); print(open("gpurun_out/bench_c2_$TAG.err").read()[-1200:])
PY
done
timeout 600 python bench.py --config 2 --no-cpu-baseline --no-side-runs --persistent 0 --kernel a1_il > gpurun_out/bench_c2_np_$TAG.json 2>/dev/null; python -c "
import json; j=json.loads(open('gpurun_out/bench_c2_np_$TAG.json').read().strip().splitlines()[-1]); print('p2 one launch per step', j['value'], j['config']['kernel'], 'resident', j['resident_launch']['value'])"
