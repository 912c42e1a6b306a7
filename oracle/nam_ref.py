"""ctypes front end of oracle/_ref/libnam_ref.so: the reference's OWN, unmodified C++ sources (read from
/root/reference at build time, never copied) compiled against the self-written Eigen stand-in in
oracle/eigen_shim. TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline): it pins the oracle to the
reference's control flow; it is never on the product path.

The library can only be built where /root/reference exists (this container); the GPU box uses the prebuilt
.so that travels with the repository snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libnam_ref.so")
# the reference's DEFAULT build (NAM_ENABLE_A2_FAST on, CMakeLists.txt:58): A2-shaped files run wavenet/a2_fast.cpp
LIB_A2FAST = os.path.join(HERE, "_ref", "libnam_ref_a2fast.so")
REFERENCE = "/root/reference"
_L = {}


def available() -> bool:
    return (os.path.exists(LIB) and os.path.exists(LIB_A2FAST)) or os.path.isdir(os.path.join(REFERENCE, "NAM"))


def build(force: bool = False, a2_fast: bool = False) -> str:
    """Compile the reference where it lies (needs /root/reference); no-op when the .so files are already there."""
    want = LIB_A2FAST if a2_fast else LIB
    if os.path.exists(LIB) and os.path.exists(LIB_A2FAST) and not force:
        return want
    if not os.path.isdir(os.path.join(REFERENCE, "NAM")):
        raise RuntimeError("oracle/_ref cannot be built here: /root/reference is absent and no prebuilt libnam_ref*.so travelled along")
    subprocess.run(["make", "-f", os.path.join(HERE, "Makefile.ref"), "-j4"], check=True, capture_output=True)
    return want


def lib(a2_fast: bool = False):
    if a2_fast not in _L:
        L = ctypes.CDLL(build(a2_fast=a2_fast))
        vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.ref_load.restype = vp
        L.ref_load.argtypes = [ctypes.c_char_p, ci, ctypes.c_char_p, ci]
        L.ref_load_lut.restype = vp
        L.ref_load_lut.argtypes = [ctypes.c_char_p, ci, ctypes.c_char_p, ctypes.c_float, ctypes.c_float, ci, ctypes.c_char_p, ci]
        L.ref_free.argtypes = [vp]
        for f in ("ref_in_channels", "ref_out_channels", "ref_prewarm_samples"):
            getattr(L, f).restype = ci
            getattr(L, f).argtypes = [vp]
        L.ref_expected_sample_rate.restype = cd
        L.ref_expected_sample_rate.argtypes = [vp]
        L.ref_set_slimmable.restype = ci
        L.ref_set_slimmable.argtypes = [vp, cd]
        L.ref_reset.restype = ci
        L.ref_reset.argtypes = [vp, cd, ci]
        L.ref_last_error.restype = ctypes.c_char_p
        L.ref_process.restype = ci
        L.ref_process.argtypes = [vp, vp, vp, ctypes.c_long, ci]
        _L[a2_fast] = L
    return _L[a2_fast]


class RefDSP:
    """nam::get_dsp(path) of the reference; planar float32 numpy I/O like the oracle's classes."""

    def __init__(self, path: str, fast_tanh: bool = False, a2_fast: bool = False, lut=None):
        """a2_fast: use the reference's default build, where A2-shaped WaveNets run wavenet/a2_fast.cpp.
        lut = (name, min, max, n_points): Activation::enable_lut(...) is in force while the model is constructed."""
        self._L = lib(a2_fast)
        err = ctypes.create_string_buffer(512)
        if lut is None:
            self._h = self._L.ref_load(path.encode(), 1 if fast_tanh else 0, err, 512)
        else:
            self._h = self._L.ref_load_lut(path.encode(), 1 if fast_tanh else 0, lut[0].encode(), float(lut[1]),
                                           float(lut[2]), int(lut[3]), err, 512)
        if not self._h:
            raise RuntimeError(err.value.decode(errors="replace"))
        self.max_buffer_size = 0

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.ref_free(self._h)
            self._h = None

    def NumInputChannels(self):
        return self._L.ref_in_channels(self._h)

    def NumOutputChannels(self):
        return self._L.ref_out_channels(self._h)

    def GetPrewarmSamples(self):
        return self._L.ref_prewarm_samples(self._h)

    def GetExpectedSampleRate(self):
        return self._L.ref_expected_sample_rate(self._h)

    def SetSlimmableSize(self, val: float):
        if self._L.ref_set_slimmable(self._h, float(val)) != 0:
            raise RuntimeError("not a SlimmableModel")

    def Reset(self, sample_rate: float, max_buffer_size: int):
        if self._L.ref_reset(self._h, float(sample_rate), int(max_buffer_size)) != 0:
            raise RuntimeError("Reset failed: " + (self._L.ref_last_error() or b"").decode(errors="replace"))
        self.max_buffer_size = int(max_buffer_size)

    def process_stream(self, x: np.ndarray, block: int) -> np.ndarray:
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float32)
        out = np.zeros((self.NumOutputChannels(), x.shape[1]), dtype=np.float32)
        if self._L.ref_process(self._h, x.ctypes.data, out.ctypes.data, x.shape[1], int(block)) != 0:
            raise RuntimeError("process failed (Reset first; block <= max buffer size)")
        return out


def get_dsp(path: str, fast_tanh: bool = False, a2_fast: bool = False, lut=None) -> RefDSP:
    return RefDSP(path, fast_tanh, a2_fast, lut)
