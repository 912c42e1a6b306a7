"""Developer probe: nam_a1_q_kernel against nam_a1_p2_kernel (NAM_HIP_NO_PIPE=1) on the same calls, block by block,
stream by stream — plain launches (whole blocks, a ragged tail, split calls that hand the state back and forth between
the two kernels) and persistent sessions; then timing of a resident launch against nam_a1_p4_kernel (NAM_HIP_A1Q=0).
Usage: a1q_probe.py [n_streams] [--time]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import neuralampmodelercore_amd as nam
from signals import stream_bank

path = os.path.join(ROOT, "tests", "golden", "models", "wavenet_a1_standard.nam")
n_streams = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
NB = 24
n = 64 * NB
x = stream_bank(n_streams, n, seed=7)


def batch(which, max_frames, ns=None):
    os.environ["NAM_HIP_NO_PIPE"] = "1" if which == "p2" else "0"
    os.environ["NAM_HIP_A1Q"] = "0" if which == "p4" else "1"
    model = nam.get_dsp(path, fast_tanh=True)
    b = model.batch(ns or n_streams, max_frames)
    b.set_kernel(nam.KERNEL_A1_IL)
    return b


def run(which, prewarm, mode, cuts=None):
    b = batch(which, 64 if mode == "persistent" else n)
    b.Reset(prewarm=prewarm)
    xd = torch.from_numpy(x[:, None, :]).cuda()
    yd = torch.zeros_like(xd)
    torch.cuda.synchronize()
    if mode == "persistent":
        assert b.set_persistent(True)
        name = b.kernel_name()
        for k in range(NB):
            b.process_device(xd.data_ptr() + k * 256, yd.data_ptr() + k * 256, 64, n)
            if k in (0, 5):
                b.flush()
        b.flush()
    else:
        name = b.kernel_name(n)
        for a_, z_ in cuts:
            b.process_device(xd.data_ptr() + a_ * 4, yd.data_ptr() + a_ * 4, z_ - a_, n)
    b.synchronize()
    torch.cuda.synchronize()
    y = yd.cpu().numpy()[:, 0, :].copy()
    b.close()
    return y, name


def report(tag, ya, yb, upto=n):
    d = np.abs(ya[:, :upto] - yb[:, :upto])
    per_block = d[:, :upto // 64 * 64].reshape(n_streams, -1, 64).max(axis=2)
    bad = d > 1e-4
    print(f"{tag}: max |q - p2| = {d.max():.2e}; per block (worst stream): " + " ".join(f"{v:.0e}" for v in per_block.max(axis=0)), flush=True)
    if bad.any():
        for s in range(min(n_streams, 4)):
            if bad[s].any():
                f = int(np.argmax(bad[s]))
                print(f"    stream {s}: first bad frame {f} (block {f // 64}, frame {f % 64}); q {ya[s, f:f + 4]} p2 {yb[s, f:f + 4]}")
    return not bad.any()


ok = True
if "--time" not in sys.argv:
    for prewarm in (False, True):
        whole = [(0, n)]
        y2, n2 = run("p2", prewarm, "launch", whole)
        yq, nq = run("q", prewarm, "launch", whole)
        print(f"kernels: {nq} vs {n2}")
        ok &= report(f"prewarm={prewarm} one launch of {NB} blocks", yq, y2)
        ragged = [(0, 128), (128, 129), (129, 400), (400, 464), (464, n - 35)]
        y2, _ = run("p2", prewarm, "launch", ragged)
        yq, _ = run("q", prewarm, "launch", ragged)
        ok &= report(f"prewarm={prewarm} split calls {ragged}", yq, y2, n - 35)
        yq, nq = run("q", prewarm, "persistent")
        y2, _ = run("p2", prewarm, "launch", whole)
        ok &= report(f"prewarm={prewarm} persistent session ({nq})", yq, y2)
    print("A1Q PROBE", "OK" if ok else "FAILED", flush=True)

# timing: one resident launch walking many buffers (what a session does between two pauses)
steps = 600
xl = torch.from_numpy(stream_bank(256, 64 * steps, seed=3)[:, None, :]).cuda()
for which in ("p4", "q", "p4", "q"):
    b = batch(which, 64 * steps, 256)
    b.Reset(prewarm=True)
    yl = torch.zeros_like(xl)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        b.process_device(xl.data_ptr(), yl.data_ptr(), 64 * steps, 64 * steps)
        b.synchronize()
        dt = time.perf_counter() - t0
    print(f"resident launch, 256 streams x {steps} buffers, {b.kernel_name(64 * steps)}: {dt / steps * 1e6:.2f} us per buffer", flush=True)
    b.close()
