"""Developer tool: nam_a1_q_kernel's time per buffer — one resident launch of 600 buffers (sustained clocks), and the kernel's
own clock for plain launches of 20 and 200 buffers (stamped instantiation: prologue, first output, period, total)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import neuralampmodelercore_amd as nam
from signals import stream_bank

path = os.path.join(ROOT, "tests", "golden", "models", "wavenet_a1_standard.nam")
model = nam.get_dsp(path, fast_tanh=True)
steps = 600
xl = torch.from_numpy(stream_bank(256, 64 * steps, seed=3)[:, None, :]).cuda()
b = model.batch(256, 64 * steps)
b.set_kernel(nam.KERNEL_A1_IL)
b.Reset(prewarm=True)
yl = torch.zeros_like(xl)
torch.cuda.synchronize()
ts = []
for rep in range(4):
    t0 = time.perf_counter()
    b.process_device(xl.data_ptr(), yl.data_ptr(), 64 * steps, 64 * steps)
    b.synchronize()
    ts.append((time.perf_counter() - t0) / steps * 1e6)
out = f"{b.kernel_name(64 * steps)}: 600-buffer launches {' '.join(f'{t:.2f}' for t in ts)} us/buffer"
b.close()
for nb in (20, 200):
    b = model.batch(256, 64 * nb)
    b.set_kernel(nam.KERNEL_A1_IL)
    b.Reset(prewarm=True)
    for rep in range(3):
        tl = b.debug_timeline(64 * nb)
    b.close()
    t0 = tl[:16, 0].min()
    last = 15
    first_out, last_out, left = (tl[last, 2] - t0) / 2400, (tl[last, 3] - t0) / 2400, (tl[:16, 4].max() - t0) / 2400
    out += f" | {nb} buffers in-kernel: prologue {(tl[last, 1] - t0) / 2400:.1f}, first output {first_out:.1f}, period {(last_out - first_out) / (nb - 1):.2f}, total {left:.1f} us"
print(out, flush=True)
