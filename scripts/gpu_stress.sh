#!/bin/bash
# stress / determinism check of the MFMA kernel: the same input rendered repeatedly (block launches, one resident
# launch, different stream counts) must be bit-identical run to run and match the oracle on sampled streams
cd "$GRAFT_REPO_ROOT"
timeout 600 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, ".")
import neuralampmodelercore_amd as nam
from oracle import nam_oracle as orc
from tests.signals import stream_bank
orc.build()
p = "tests/golden/models/wavenet_a1_standard.nam"
m = nam.get_dsp(p, fast_tanh=True)
bad = 0
for n_streams, n_blocks in ((256, 120), (1024, 60), (3000, 20)):
    x = stream_bank(n_streams, 64 * n_blocks + 13, seed=123)
    outs = []
    for rep in range(4):
        b = m.batch(n_streams, 64)
        b.Reset(prewarm=True)
        y = b.process_stream(x, 64) if rep % 2 == 0 else np.concatenate(b.render(list(x)), axis=0)[:, None, :]
        outs.append(y)
        b.close()
    same_block = np.array_equal(outs[0], outs[2])
    same_res = np.array_equal(outs[1], outs[3])
    cross = float(np.max(np.abs(outs[0] - outs[1])))
    errs = []
    for s in (0, n_streams // 2, n_streams - 1):
        r = orc.get_dsp(p, fast_tanh=True)
        r.Reset(48000.0, 64)
        errs.append(float(np.max(np.abs(r.process_stream(x[s], 64) - outs[0][s]))))
    print(n_streams, "streams", n_blocks, "blocks: block-mode deterministic", same_block, "| resident deterministic", same_res,
          "| block vs resident max diff", cross, "| vs oracle", max(errs), flush=True)
    bad += (not same_block) + (not same_res) + (cross > 1e-5) + (max(errs) > 5e-5)
print("STRESS", "FAILED" if bad else "OK")
PY
