"""developer probe: what does a stream of small device-side writes on ANOTHER stream cost a resident launch?
(hipStreamWriteValue64 / 8-byte hipMemcpyAsync are small kernels; do their boundaries slow the resident kernel?)
Also: is fine-grained device memory host-writable here (a doorbell the host could store to directly)?"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "finegrained":
    hip = ctypes.CDLL("libamdhip64.so")
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(4096), ctypes.c_uint(int(sys.argv[2])))
    print("hipExtMallocWithFlags rc", rc, hex(p.value or 0), flush=True)
    v = (ctypes.c_ulonglong * 1)(0x1234)
    ctypes.memmove(p.value, v, 8)  # segfaults when the memory is not host-mapped
    back = (ctypes.c_ulonglong * 1)()
    ctypes.memmove(back, p.value, 8)
    print("host store/load ok", hex(back[0]), flush=True)
    t0 = time.perf_counter()
    for i in range(1000):
        ctypes.memmove(back, p.value, 8)
    print("host load us", (time.perf_counter() - t0) * 1e3, flush=True)
    sys.exit(0)

import torch
import neuralampmodelercore_amd as nam
from signals import stream_bank
hip = ctypes.CDLL("libamdhip64.so")
hip.hipStreamWriteValue64.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
model = nam.get_dsp(os.path.join(ROOT, "tests/golden/models/wavenet_a1_standard.nam"), fast_tanh=True)
n, nb = 256, 2000
x = torch.from_numpy(stream_bank(n, 64 * nb, seed=1)[:, None, :]).cuda()
y = torch.zeros_like(x)
b = model.batch(n, 64)
b.set_kernel(nam.KERNEL_A1_IL)
b.Reset(prewarm=True)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
dummy = torch.zeros(64, dtype=torch.int64, device="cuda")
pinned = torch.zeros(64, dtype=torch.int64).pin_memory()
torch.cuda.synchronize()

def run(mode, gap_us):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(sa)
    b.process_device(x.data_ptr(), y.data_ptr(), 64 * nb, x.shape[2], sa.cuda_stream)
    e1.record(sa)
    k = 0
    while not e1.query():
        if mode == "write":
            hip.hipStreamWriteValue64(sb.cuda_stream, dummy.data_ptr(), k, 0)
        elif mode == "memcpy":
            hip.hipMemcpyAsync(dummy.data_ptr(), pinned.data_ptr(), 8, 1, sb.cuda_stream)
        elif mode == "kernel":
            with torch.cuda.stream(sb):
                dummy[:1].add_(1)
        k += 1
        t = time.perf_counter() + gap_us * 1e-6
        while time.perf_counter() < t:
            pass
    torch.cuda.synchronize()
    print(f"{mode:8s} gap {gap_us:5.1f} us: {e0.elapsed_time(e1) * 1e3 / nb:7.3f} us per block, {k} side operations", flush=True)

for mode, gap in [("none", 5), ("none", 5), ("write", 8), ("write", 3), ("write", 0), ("memcpy", 3), ("kernel", 3), ("none", 5)]:
    run(mode, gap)
b.close()
for flags in (1, 2, 3):
    r = subprocess.run([sys.executable, __file__, "finegrained", str(flags)], capture_output=True, text=True, timeout=60)
    print("finegrained flags", flags, "rc", r.returncode, r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-200:])
