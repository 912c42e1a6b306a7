"""developer probe: persistent block mode, time from the first command to host-visible results for K buffers,
as a function of K and of what the GPU did just before (idle gap / device synchronize)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import neuralampmodelercore_amd as nam
from signals import stream_bank
model = nam.get_dsp(os.path.join(ROOT, "tests/golden/models/wavenet_a1_standard.nam"), fast_tanh=True)
n = 256
x = torch.from_numpy(stream_bank(n, 64 * 400, seed=1)[:, None, :]).cuda()
y = torch.zeros_like(x)
b = model.batch(n, 64)
assert b.set_persistent(True)
b.Reset(prewarm=True)
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
torch.cuda.synchronize()

def region(K, pre):
    if pre == "sync":
        torch.cuda.synchronize()
    elif pre == "idle1ms":
        torch.cuda.synchronize(); t = time.perf_counter() + 1e-3
        while time.perf_counter() < t: pass
    t0 = time.perf_counter()
    for k in range(K):
        b.process_device(x.data_ptr() + k * 256, y.data_ptr() + k * 256, 64, x.shape[2], st.cuda_stream)
    t1 = time.perf_counter()
    b.flush(st.cuda_stream)
    t2 = time.perf_counter()
    return (t1 - t0) * 1e6, (t2 - t0) * 1e6

def bench_like(K, W, pitch_frames):
    xs = x[:, :, :pitch_frames].contiguous(); ys = torch.zeros_like(xs)
    out = []
    for rep in range(11):
        for k in range(W):
            b.process_device(xs.data_ptr() + k * 256, ys.data_ptr() + k * 256, 64, pitch_frames, st.cuda_stream)
        b.flush(st.cuda_stream); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(W, W + K):
            b.process_device(xs.data_ptr() + k * 256, ys.data_ptr() + k * 256, 64, pitch_frames, st.cuda_stream)
        b.flush(st.cuda_stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    out.sort()
    return out[5]
for W, pitch in ((5, 1600), (0, 1600), (5, 25600), (0, 25600), (5, 1600)):
    v, w_ = bench_like(20, W, pitch)
    print(f"bench-like K=20 W={W} pitch={pitch}: visible {v:7.1f} us, after device sync {w_:7.1f} us", flush=True)
for pre in ("sync",):
    for K in (1, 2, 3, 5, 10, 20, 40, 80, 160, 320):
        r = sorted(region(K, pre) for _ in range(9))
        enq, vis = r[4]
        print(f"pre={pre:8s} K={K:4d} enqueue {enq:7.1f} us  visible {vis:8.1f} us  per-buffer {vis / K:6.2f}  min {r[0][1]:8.1f}", flush=True)
b.close()
