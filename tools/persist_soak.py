"""Developer tool: a long persistent session of a model (default: the headline model) — N streams x B buffers, repeated R
times with random pauses (the launch leaves and restarts) — against the same audio rendered by ONE ordinary multi-block
launch of the un-pipelined kernel of the family (NAM_HIP_MAX_STAGES=1 in a fresh batch); every stream, every frame.
Usage: persist_soak.py streams buffers reps [model fixture name]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import neuralampmodelercore_amd as nam
from signals import stream_bank

n_streams, nb, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
block = 64
name = sys.argv[4] if len(sys.argv) > 4 else "wavenet_a1_standard"
model = nam.get_dsp(os.path.join(ROOT, f"tests/golden/models/{name}.nam"), fast_tanh=True)
x = stream_bank(n_streams, nb * block, seed=99)
xd = torch.from_numpy(x[:, None, :]).cuda()
os.environ["NAM_HIP_MAX_STAGES"] = "1"
ref_b = model.batch(n_streams, nb * block)
ref_b.Reset(prewarm=True)
yr = torch.zeros_like(xd)
ref_b.process_device(xd.data_ptr(), yr.data_ptr(), nb * block, nb * block)
ref_b.synchronize()
ref_b.close()
os.environ["NAM_HIP_MAX_STAGES"] = "0"
rng = np.random.default_rng(5)
worst = 0.0
for rep in range(reps):
    b = model.batch(n_streams, block)
    assert b.set_persistent(True)
    b.Reset(prewarm=True)
    yd = torch.zeros_like(xd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(nb):
        b.process_device(xd.data_ptr() + k * block * 4, yd.data_ptr() + k * block * 4, block, nb * block)
        if rep and rng.integers(0, 400) == 0:
            time.sleep(0.0005)
    b.flush()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    err = float((yd - yr).abs().max())
    bad = int(((yd - yr).abs() > 1e-4).any(dim=2).sum())
    worst = max(worst, err)
    if bad:
        d = (yd - yr).abs()[:, 0, :] > 1e-4
        ids = torch.nonzero(d.any(dim=1))[:, 0].cpu().numpy()
        first = [int(torch.nonzero(d[i])[0, 0]) for i in ids[:12]]
        print(f"      bad streams {ids[:12].tolist()} ... {ids[-4:].tolist()}; first bad frame of each: {first} (buffer {[f // block for f in first]})", flush=True)
    print(f"   soak rep {rep}: {b.kernel_name()} {nb} buffers in {dt * 1e3:.1f} ms ({dt / nb * 1e6:.2f} us/buffer), max |session - one launch| = {err:.3e}, bad streams {bad}", flush=True)
    b.close()
print("   SOAK", "OK" if worst < 1e-4 else "FAILED")
