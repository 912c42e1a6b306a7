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
bad = 0
CASES = [("wavenet_a1_standard", nam.KERNEL_AUTO, (256, 120)), ("wavenet_a1_standard", nam.KERNEL_AUTO, (1024, 60)),
         ("wavenet_a1_standard", nam.KERNEL_AUTO, (3000, 20)),
         # the K-tap MFMA kernel (forced: AUTO leaves it above 1,024 streams), long runs: ring rows written by one
         # wavefront are read by the others blocks later
         ("A2", nam.KERNEL_A1_MFMA, (256, 150)), ("A2", nam.KERNEL_A1_MFMA, (1500, 30)), ("synth_kt_c16", nam.KERNEL_A1_MFMA, (700, 40))]
for name, kernel, (n_streams, n_blocks) in CASES:
    p = f"tests/golden/models/{name}.nam"
    m = nam.get_dsp(p, fast_tanh=True)
    x = stream_bank(n_streams, 64 * n_blocks + 13, seed=123)
    outs = []
    for rep in range(4):
        b = m.batch(n_streams, 64)
        b.set_kernel(kernel)
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
    print(name, n_streams, "streams", n_blocks, "blocks: block-mode deterministic", same_block, "| resident deterministic", same_res,
          "| block vs resident max diff", cross, "| vs oracle", max(errs), flush=True)
    bad += (not same_block) + (not same_res) + (cross > 1e-5) + (max(errs) > 5e-5)
print("STRESS", "FAILED" if bad else "OK")
PY
