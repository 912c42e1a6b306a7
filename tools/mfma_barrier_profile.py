"""Developer tool: how long each wave of workgroup 0 of nam_a1_mfma_kernel sits in barriers (shader cycles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neuralampmodelercore_amd as nam

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 64 * 50
m = nam.get_dsp(os.path.join(ROOT, "tests/golden/models/wavenet_a1_standard.nam"), fast_tanh=True)
b = m.batch(streams, 64)
b.set_kernel(nam.KERNEL_A1_MFMA)
b.Reset(prewarm=True)
t = b.debug_timeline(nfr)
jobs = (nfr + 63) // 64 * 20
for w in range(8):
    bar, tot = int(t[w, 0]), int(t[w, 1])
    print(f"wave {w} ({'compute' if w < 4 else 'mover'}): total {tot} cyc = {tot / jobs:.0f}/job, in barrier {bar} = {bar / jobs:.0f}/job ({100.0 * bar / max(tot, 1):.0f}%)")
    if w < 4:
        names = ["barrier exit->taps ready", "->conv done", "->x updated", "->published", "->job end (bookkeeping)"]
        print("      " + "; ".join(f"{n} {int(t[w, 2 + k]) / jobs:.0f}" for k, n in enumerate(names)))
