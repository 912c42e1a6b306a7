"""Developer probe: the four official NAM WaveNet sizes (standard 16->8, lite 12->6, feather 8->4, nano 4->2; ten layers
per array, dilations 1..512, kernel size 3) as seeded random models: which kernel AUTO picks and what it sustains."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import neuralampmodelercore_amd as nam
import make_synthetic_models as msm
from signals import stream_bank

D = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
SIZES = {"standard": (16, 8), "lite": (12, 6), "feather": (8, 4), "nano": (4, 2)}
n, block, K = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 64, 400
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as tmp:
    os.makedirs(os.path.join(tmp, "models"))
    msm.HERE = tmp
    for name, (c0, c1) in SIZES.items():
        msm.build("size_" + name, [(c0, D, "Tanh", False), (c1, D, "Tanh", True)], 5)
        m = nam.get_dsp(os.path.join(tmp, "models", "size_" + name + ".nam"), fast_tanh=True)
        T = block * K
        x = torch.from_numpy(stream_bank(n, T, seed=1)[:, None, :]).to(dev)
        y = torch.zeros_like(x)
        for pers in (False, True):
            b = m.batch(n, block)
            b.Reset(prewarm=True)
            if pers and not b.set_persistent(True):
                b.close()
                continue
            st = torch.cuda.Stream(dev); sh = st.cuda_stream
            def run():
                for s in range(K):
                    b.process_device(x.data_ptr() + s * block * 4, y.data_ptr() + s * block * 4, block, T, sh)
                b.flush(sh)
            run(); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
            print(f"{name:9s} {c0:2d}->{c1:d}  {b.kernel_name():20s} {'persistent block mode' if pers else 'one launch per block '} {n} streams: "
                  f"{dt*1e6:7.2f} us per 64-frame block = {n*block/48000/dt:9.0f} xRT", flush=True)
            b.close()
