"""developer probe: at which stream counts does the persistent block mode keep up? (one workgroup per stream: the
resident launch needs every workgroup on the chip at once)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import neuralampmodelercore_amd as nam
from signals import stream_bank
model = nam.get_dsp(os.path.join(ROOT, "tests/golden/models/wavenet_a1_standard.nam"), fast_tanh=True)
for n in [int(a) for a in sys.argv[1:]] or [7, 64, 128, 200, 240, 248, 255, 256]:
    x = torch.from_numpy(stream_bank(n, 64 * 40, seed=1)[:, None, :]).cuda()
    y = torch.zeros_like(x)
    b = model.batch(n, 64)
    ok = b.set_persistent(True)
    b.Reset(prewarm=True)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    t0 = time.perf_counter()
    try:
        for k in range(40):
            b.process_device(x.data_ptr() + k * 256, y.data_ptr() + k * 256, 64, x.shape[2], st.cuda_stream)
        b.flush(st.cuda_stream)
        dt = time.perf_counter() - t0
        print(n, "eligible", ok, "ok", f"{dt * 1e6 / 40:.1f} us per buffer", "finite", bool(torch.isfinite(y).all()), flush=True)
    except Exception as e:
        print(n, "eligible", ok, "FAILED", str(e)[:120], f"{time.perf_counter() - t0:.2f} s", flush=True)
    try:
        b.close()
    except Exception as e:
        print("close failed", str(e)[:100])
