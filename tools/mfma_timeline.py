"""Developer tool: per-job phase breakdown of the MFMA kernel (workgroup 0) in shader cycles."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neuralampmodelercore_amd as nam

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = nam.get_dsp(os.path.join(ROOT, "tests/golden/models/wavenet_a1_standard.nam"), fast_tanh=True)
b = m.batch(streams, 64)
b.set_kernel(nam.KERNEL_A1_MFMA)
b.Reset(prewarm=True)
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 64 * 4
t = b.debug_timeline(nfr)
names = ["barrier", "opreads", "stash", "fetch", "conv", "act+1x1", "publish"]
print("job  start   " + "  ".join(f"{n:>8}" for n in names) + "    total")
t0 = t[0, 0]
for j in range(94):
    if t[j, 0] == 0:
        break
    r = t[j]
    # stamps: 0 job start, 1 after barrier, 6 operand reads landed, 7 after ring store + stash, 2 after fetch issue,
    #         3 after conv MFMAs (LAYER), 4 after act + 1x1 (LAYER), 5 end of job
    ph = [r[1] - r[0], r[6] - r[1], r[7] - r[6], r[2] - r[7], (r[3] - r[2]) if r[3] else 0, (r[4] - r[3]) if r[4] else 0,
          r[5] - (r[4] if r[4] else r[2])]
    nxt = t[j + 1, 0] if j + 1 < 94 and t[j + 1, 0] else r[5]
    print(f"{j:3d} {r[0]-t0:7d}   " + "  ".join(f"{p:8d}" for p in ph) + f"  {nxt - r[0]:7d}")

for row, name in ((95, "first WG"), (94, "last WG")):
    e, l, x = t[row, 0], t[row, 1], t[row, 2]
    print(f"{name}: entry->loop {(l - e) * 10} ns, loop->exit {(x - l) * 10} ns, entry offset vs first WG {(e - t[95, 0]) * 10} ns, exit vs first entry {(x - t[95, 0]) * 10} ns")
