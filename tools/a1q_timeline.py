"""Developer tool: per-stage timeline of nam_a1_q_kernel's stamped instantiation (nam_hip_batch_debug_timeline): one plain
launch of N buffers of silence, 256 streams; workgroup 0's twelve stages report shader-clock stamps.
Usage: a1q_timeline.py [buffers ...]   (default 20 100)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import neuralampmodelercore_amd as nam

path = os.path.join(ROOT, "tests", "golden", "models", "wavenet_a1_standard.nam")
model = nam.get_dsp(path, fast_tanh=True)
names = ["L0", "L1", "L2", "L3", "L4", "L5", "L6", "L7", "L8", "L9", "T+M0", "M1-2", "M3-4", "M5-6", "M7-8", "M9+head"]
simd = ["A", "A", "A", "B", "B", "B", "C", "C", "D", "D", "B", "C", "C", "D", "D", "A"]
for nb in [int(a) for a in sys.argv[1:]] or [20, 100]:
    b = model.batch(256, 64 * nb)
    b.set_kernel(nam.KERNEL_A1_IL)
    b.Reset(prewarm=True)
    for rep in range(3):
        t0 = time.perf_counter()
        tl = b.debug_timeline(64 * nb)
        dt = time.perf_counter() - t0
    t_entry = tl[:16, 0].min()
    print(f"== {nb} buffers, {b.kernel_name(64 * nb)}; host round trip {dt * 1e6:.0f} us; cycles from the first wave's entry (2.4 GHz -> us)")
    print("stage      SIMD  entry  prologue_done  first_handover  last_handover  left   |  wait_in  wait_out  busy/unit  units")
    for s in range(16):
        e, p, f, l, x, wi, wo, n = [int(v) for v in tl[s]]
        units = max(n, 1)
        span = l - p
        print(f"{names[s]:10s} {simd[s]:4s} {(e - t_entry) / 2400:6.2f} {(p - t_entry) / 2400:13.2f} {(f - t_entry) / 2400:15.2f} {(l - t_entry) / 2400:14.2f} {(x - t_entry) / 2400:6.2f}"
              f"   | {wi / 2400:8.2f} {wo / 2400:9.2f} {(span - wi - wo) / units / 2400:10.3f} {n:6d}")
    b.close()
