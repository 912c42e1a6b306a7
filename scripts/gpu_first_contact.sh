#!/bin/bash
# developer tool: first-contact check of a kernel variant under a hard timeout (a hung barrier must not eat the box)
cd "$GRAFT_REPO_ROOT"
K=${1:-a1_mfma}
timeout 120 python - <<PY
import numpy as np, sys, os
sys.path.insert(0, ".")
import neuralampmodelercore_amd as nam
from oracle import nam_oracle as orc
from tests.signals import stream_bank
orc.build()
kern = {"a1_mfma": nam.KERNEL_A1_MFMA}["$K"]
for name in ("wavenet_a1_standard", "wavenet", "slimmable_wavenet"):
    p = os.path.join("tests/golden/models", name + ".nam")
    m = nam.get_dsp(p, fast_tanh=True)
    print(name, "has", m.info.has_a1_kernel, flush=True)
    if not (m.info.has_a1_kernel & 4):
        continue
    for n in (64, 64 * 3 + 5):
        x = stream_bank(3, n, seed=3)
        b = m.batch(3, 64)
        b.set_kernel(kern)
        b.Reset(prewarm=True)
        y = b.process_stream(x, 64)
        r = orc.get_dsp(p, fast_tanh=True)
        r.Reset(48000, 64)
        ref = r.process_stream(x[1], 64) if hasattr(r, "process_stream") else None
        print(name, n, "err", float(np.max(np.abs(ref - y[1]))) if ref is not None else "n/a", flush=True)
        b.close()
PY
echo "exit $?"
