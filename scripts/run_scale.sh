#!/bin/bash
# The multi-GPU records this repo cannot make from its 1-GPU build box: bench.py for the headline config (2) and the
# 8-GPU config (4: wavenet_a2_max.nam, 4,096 streams = 512 per GPU) on N GPUs of one node, one process per GPU, plus
# the C++ tool's thread-per-device render. Streams never interact: ranks share no data-path collective (RCCL only
# broadcasts the model text, scatters the input bank and gathers the rendered tail, neuralampmodelercore_amd/sharding.py).
#   bash scripts/run_scale.sh [N=8] [out_dir=profiles/scale]      (STEPS / WARMUP default to the driver's own 20 / 5: the N = 1 line is the BENCH line)
# Writes <out_dir>/bench_config{2,4}_n<N>.json (the bench's one JSON line) and render_devices_n<N>.txt.
cd "$(dirname "$0")/.."
N=${1:-8}; OUT=${2:-profiles/scale}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for C in 2 4; do
  if [ "$N" -gt 1 ]; then
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29517} \
      bench.py --gpus "$N" --config $C --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-other-configs | grep '^{' > "$OUT/bench_config${C}_n${N}.json"
  else
    python bench.py --gpus 1 --config $C --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-other-configs | grep '^{' > "$OUT/bench_config${C}_n${N}.json"
  fi
  python - "$OUT/bench_config${C}_n${N}.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"config {j['config']['baseline_config']}: {j['n_gpus']} GPUs, {j['value']:,.0f} {j['unit'].split(' ')[0]}, {j['ms_per_step'] * 1e3:.2f} us/step, kernel {j['config']['kernel']}")
print(f"  parity vs oracle, every rank's first and last stream (global streams {j.get('parity_streams_checked')}): per rank {j.get('parity_per_rank')}, max {j.get('max_abs_err_vs_oracle')}")
PY
done
# the C++ tool: 8 x N short files dealt to N devices, one batch + host thread per device (cpp/NAM/multi_device.h)
if [ -x cpp/tools/render ]; then
  T=$(mktemp -d)
  python - "$T" "$N" <<'PY'
import struct, sys, numpy as np
d, n = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(1)
for i in range(8 * n):  # mono float32 WAV, 48 kHz, ragged lengths
    data = (0.2 * rng.standard_normal(48000 * 2 + 997 * i)).astype("<f4").tobytes()
    open(f"{d}/in{i:03d}.wav", "wb").write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, 48000, 48000 * 4, 4, 32)
                                        + b"data" + struct.pack("<I", len(data)) + data)
PY
  /usr/bin/time -f "render --devices 0-$((N-1)): %e s wall for $((8*N)) files" \
    cpp/tools/render --devices 0-$((N-1)) tests/golden/models/wavenet_a1_standard.nam --batch "$T/out" "$T"/in*.wav 2> "$OUT/render_devices_n${N}.txt"
  tail -1 "$OUT/render_devices_n${N}.txt"
  rm -rf "$T"
fi
