"""Developer probe: A2-Lite (A2.nam at ratio 0.2: 3 channels) — which kernel serves it best at which stream count.
   python tools/a2_lite_probe.py   (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import neuralampmodelercore_amd as nam

model = nam.get_dsp(os.path.join(ROOT, "tests/golden/models/A2.nam"), fast_tanh=True)
block, nb = 64, 400
for n_streams in (256, 512, 1024, 2048):
    xd = torch.zeros((n_streams, 1, 64 * block), dtype=torch.float32, device="cuda")
    yd = torch.zeros_like(xd)
    for kname, k in (("auto", nam.KERNEL_AUTO), ("a1 (VALU)", nam.KERNEL_A1), ("wn_reg", nam.KERNEL_WN_REG)):
        for pers in (True, False):
            b = model.batch(n_streams, block)
            b.set_kernel(k)
            b.Reset(prewarm=True)
            b.SetSlimmableSize(0.2)
            on = b.set_persistent(True) if pers else False
            if pers and not on:
                b.close()
                continue
            def run(n):
                for i in range(n):
                    off = (i % 64) * block * 4
                    b.process_device(xd.data_ptr() + off, yd.data_ptr() + off, block, 64 * block)
                b.flush(); b.synchronize()
            run(64)
            t0 = time.perf_counter(); run(nb); dt = time.perf_counter() - t0
            print(f"A2-Lite {n_streams:5d} streams  kernel {kname:10s} {b.kernel_name():22s} sessions {str(on):5s}  {dt / nb * 1e6:7.2f} us/buffer  {n_streams * nb * block / 48000.0 / dt:9.0f} xRT", flush=True)
            b.close()
