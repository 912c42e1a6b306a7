"""Developer probe: K per-block launches replayed from one captured HIP graph vs enqueued one by one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import neuralampmodelercore_amd as nam
from signals import stream_bank

n, block, K = 256, 64, 200
m = nam.get_dsp(os.path.join(ROOT, "tests/golden/models/wavenet_a1_standard.nam"), fast_tanh=True)
dev = torch.device("cuda", 0)
T = block * K
x = torch.from_numpy(stream_bank(n, T, seed=0)[:, None, :]).to(dev)
y = torch.zeros_like(x)
b = m.batch(n, block)
b.Reset(prewarm=True)
st = torch.cuda.Stream(dev)
sh = st.cuda_stream
xp, yp = x.data_ptr(), y.data_ptr()
def run():
    for s in range(K):
        b.process_device(xp + s * block * 4, yp + s * block * 4, block, T, sh)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); t_plain = (time.perf_counter() - t0) / K
y_plain = y.clone()
g = torch.cuda.CUDAGraph()
b.Reset(prewarm=True)
with torch.cuda.stream(st):
    g.capture_begin()
    run()
    g.capture_end()
torch.cuda.synchronize()
b.Reset(prewarm=True); y.zero_()
# bring the state to the same point as before the plain timed run: one pass of K blocks
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / K
print(f"plain {t_plain*1e6:.2f} us/step, graph {t_graph*1e6:.2f} us/step, same output {torch.equal(y, y_plain)}")
for _ in range(3):
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); print(f"  graph replay {(time.perf_counter() - t0) / K*1e6:.2f} us/step")
