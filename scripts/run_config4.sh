#!/bin/bash
# BASELINE.json configs[3]: wavenet_a2_max.nam + FiLM conditioning, 4,096 streams sharded over 8 MI355X (512 per GPU).
# One process per GPU over RCCL; streams never interact, so RCCL only broadcasts the model text, scatters the input bank
# and gathers the rendered tail (neuralampmodelercore_amd/sharding.py). Unmeasured until a SCALE record exists.
cd "$(dirname "$0")/.."
N=${1:-8}
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29517} \
  bench.py --gpus $N --config 4 --steps ${STEPS:-2000} --warmup ${WARMUP:-200}
