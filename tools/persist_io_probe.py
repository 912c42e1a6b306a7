"""Where may a persistent session's buffers live? Times one 64-frame buffer (doorbell -> flush) with the input in plain
device memory / fine-grained device memory (what a host writes through the PCIe BAR) / host-mapped memory, and the
output in device / host-mapped memory. GPU box only.  python tools/persist_io_probe.py <model.nam> [streams] [frames]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuralampmodelercore_amd as nam

model0 = nam.get_dsp(sys.argv[1], fast_tanh=True)  # (loads libnam_hip.so and, with it, the HIP runtime it is linked to)
hip_path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
hip = ctypes.CDLL(hip_path)  # the SAME runtime image, not a second copy
path = sys.argv[1]
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 256
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 64


def alloc(kind, nbytes):
    p = ctypes.c_void_p()
    if kind == "dev":
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes))
    elif kind == "bar":
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(0x1))  # hipDeviceMallocFinegrained
    else:
        rc = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(0x2 | 0x40000000))  # mapped | coherent
    assert rc == 0, (kind, rc)
    hip.hipMemset(p, 0, ctypes.c_size_t(nbytes))
    hip.hipDeviceSynchronize()
    return p.value


def upload(dst, arr):
    assert hip.hipMemcpy(ctypes.c_void_p(dst), arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(arr.nbytes), 1) == 0


def download(src, n):
    arr = np.empty(n, np.float32)
    assert hip.hipMemcpy(arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(src), ctypes.c_size_t(arr.nbytes), 2) == 0
    return arr


signal = (0.3 * np.sin(np.arange(streams * frames) * 0.013)).astype(np.float32)
first = None


model = nam.get_dsp(path, fast_tanh=True)
for kin in ("dev", "bar", "host"):
    for kout in ("dev", "host"):
        b = model.batch(streams, frames)
        b.set_persistent(True)
        b.Reset(prewarm=True)
        pin, pout = alloc(kin, streams * frames * 4), alloc(kout, streams * frames * 4)
        upload(pin, signal)
        ts = []
        for i in range(300):
            t0 = time.perf_counter()
            b.process_device(pin, pout, frames, frames)
            b.flush()
            ts.append((time.perf_counter() - t0) * 1e6)
        ts = np.sort(np.array(ts[50:]))
        print(f"{os.path.basename(path)} {b.kernel_name(frames)} streams {streams} frames {frames} in={kin} out={kout}: "
              f"p50 {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f}  p99 {ts[int(len(ts) * 0.99)]:.1f}", flush=True)
        # what the caller reads the moment flush returns: host memory directly, device memory through a copy
        seq = []
        for i in range(20):
            b.process_device(pin, pout, frames, frames)
            b.flush()
            if kout == "host":
                seq.append(np.ctypeslib.as_array(ctypes.cast(pout, ctypes.POINTER(ctypes.c_float)), (streams * frames,)).copy())
            else:
                seq.append(download(pout, streams * frames))
        y = np.stack(seq)
        if first is None:
            first = y
        print("    max |y - y(dev, dev)| over 20 buffers =", float(np.max(np.abs(y - first))), " max |y| =", float(np.max(np.abs(y))), flush=True)
        b.close()
