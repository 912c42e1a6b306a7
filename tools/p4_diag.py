"""Developer tool: where nam_a1_p4_kernel's output first departs from nam_a1_p2_kernel's (per 64-frame block)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import neuralampmodelercore_amd as nam
from signals import stream_bank

n_streams, nb = int(sys.argv[1]), int(sys.argv[2])
model = nam.get_dsp(os.path.join(ROOT, "tests/golden/models/wavenet_a1_standard.nam"), fast_tanh=True)
x = stream_bank(n_streams, nb * 64, seed=99)
outs = {}
for tag, env in (("p2", "1"), ("p4", "0")):
    os.environ["NAM_HIP_NO_PIPE"] = env
    b = model.batch(n_streams, nb * 64)
    b.set_kernel(nam.KERNEL_A1_IL)
    b.Reset(prewarm=False)
    outs[tag] = b.process(x)[:, 0, :]
    print(tag, b.kernel_name(nb * 64))
    b.close()
d = np.abs(outs["p2"] - outs["p4"]).reshape(n_streams, nb, 64)
print("non-persistent, no prewarm: max err per block (stream 0):", [f"{v:.1e}" for v in d[0].max(axis=1)[:12]])
print("frames of block 0 / 1 with err > 1e-5 (stream 0):", np.nonzero(d[0, 0] > 1e-5)[0][:20], np.nonzero(d[0, 1] > 1e-5)[0][:20])
print("worst stream", int(d.max(axis=(1, 2)).argmax()), "overall", float(d.max()))
