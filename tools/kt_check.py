"""Developer tool: K-tap MFMA kernel vs the VALU kernel and the oracle on A2.nam and the synthetic fixtures."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import neuralampmodelercore_amd as nam
import nam_oracle
from signals import stream_bank

def run(name, ratio=None, blocks=6, block=64, tail=0):
    path = os.path.join(ROOT, "tests/golden/models", name + ".nam")
    model = nam.get_dsp(path, fast_tanh=True)
    n = 3
    T = blocks * block + tail
    x = stream_bank(n, T, seed=5)
    ys = {}
    for kname, k in (("valu", nam.KERNEL_A1), ("mfma", nam.KERNEL_A1_MFMA)):
        b = model.batch(n, block)
        b.set_kernel(k)
        if ratio is not None:
            b.SetSlimmableSize(ratio)
        b.Reset(prewarm=True)
        ys[kname] = (b.process_stream(x, block), b.get_kernel())
        b.close()
    ref = nam_oracle.get_dsp(path, fast_tanh=True)
    if ratio is not None:
        ref.SetSlimmableSize(ratio)
    ref.Reset(48000.0, block)
    r = ref.process_stream(x[0], block)[0]
    for kname, (y, k) in ys.items():
        print(f"{name} ratio={ratio} {kname} (kernel id {k}): max|err| vs oracle {np.max(np.abs(y[0, 0] - r)):.3e}  "
              f"finite {np.isfinite(y).all()}", flush=True)
    print("   valu vs mfma", float(np.max(np.abs(ys['valu'][0] - ys['mfma'][0]))), flush=True)

if __name__ == "__main__":
    run("A2", 1.0)
    run("A2", 1.0, blocks=3, tail=17)
    for nm in sys.argv[1:]:
        run(nm)
