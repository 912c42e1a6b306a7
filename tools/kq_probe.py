"""developer probe: nam_kq_kernel (NAM_HIP_KQ=1) against nam_kp_kernel on the same calls, block by block."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import neuralampmodelercore_amd as nam
from signals import stream_bank

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
path = os.path.join(ROOT, "tests", "golden", "models", "A2.nam")
n_streams, n = 3, 64 * 8
x = stream_bank(n_streams, n, seed=7)


def run(kq, prewarm, persistent):
    os.environ["NAM_HIP_KQ"] = "1" if kq else "0"
    model = nam.get_dsp(path, fast_tanh=True)
    b = model.batch(n_streams, 512)
    b.Reset(prewarm=prewarm)
    xd = torch.from_numpy(x[:, None, :]).cuda()
    yd = torch.zeros_like(xd)
    torch.cuda.synchronize()
    if persistent:
        assert b.set_persistent(True)
        for k in range(n // 64):
            b.process_device(xd.data_ptr() + k * 256, yd.data_ptr() + k * 256, 64, n)
        b.flush()
    else:
        b.process_device(xd.data_ptr(), yd.data_ptr(), 256, n)
        b.process_device(xd.data_ptr() + 1024, yd.data_ptr() + 1024, 256, n)
    b.synchronize()
    torch.cuda.synchronize()
    y = yd.cpu().numpy()[:, 0, :].copy()
    b.close()
    return y


for prewarm in (False, True):
    for persistent in (False, True):
        ya, yb = run(True, prewarm, persistent), run(False, prewarm, persistent)
        d = np.abs(ya - yb).reshape(n_streams, -1, 64).max(axis=2)
        print(f"prewarm={prewarm} persistent={persistent}: max |kq - kp| per 64-frame block, stream 0: "
              + " ".join(f"{v:.1e}" for v in d[0]) + f" | kp max {np.abs(yb).max():.3f}")
        if not prewarm and not persistent:
            print("  kq[0, :8] =", ya[0, :8], "\n  kp[0, :8] =", yb[0, :8])
            e = np.abs(ya[0] - yb[0])
            print("  first frame with |diff| > 1e-5:", int(np.argmax(e > 1e-5)) if (e > 1e-5).any() else None)
