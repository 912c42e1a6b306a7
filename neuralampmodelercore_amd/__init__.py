"""
neuralampmodelercore_amd — MI355X-native NAM inference core (Python host-side binding).

The product is ``lib/libnam_hip.so`` (hand-written HIP kernels for gfx950 + the C ABI declared in
``include/nam_hip.h``).  This module is only a thin ctypes mirror of the reference's operator
surface for the hot path — ``nam::get_dsp(path)`` (NAM/get_dsp.h:85) and
``nam::DSP::process / Reset / prewarm`` (NAM/dsp.h:89-163) — batched over many independent audio
streams.  There is no CPU fallback: if the shared library cannot be loaded, importing the batch API
raises.  PyTorch is used by callers only for device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

import numpy as np

__all__ = [
    "NamHipError", "NamFileValidationError", "Model", "Batch", "get_dsp", "get_dsp_json", "get_dsp_data", "get_sample_rate_from_nam_file", "lib_path", "load_library",
    "KERNEL_AUTO", "KERNEL_GENERIC", "KERNEL_A1", "KERNEL_A1_MFMA", "KERNEL_A1_IL", "KERNEL_WN_REG", "ABI_SYMBOLS",
]

_HERE = os.path.dirname(os.path.abspath(__file__))

KERNEL_AUTO, KERNEL_GENERIC, KERNEL_A1, KERNEL_A1_MFMA, KERNEL_A1_IL, KERNEL_WN_REG = 0, 1, 2, 3, 4, 5

ERR_INVALID_ARGUMENT, ERR_FILE, ERR_MODEL, ERR_UNSUPPORTED, ERR_DEVICE, ERR_TOO_MANY_FRAMES = -1, -2, -3, -4, -5, -6

# every symbol include/nam_hip.h declares (tests check the built library exports exactly these)
ABI_SYMBOLS = [
    "nam_hip_last_error", "nam_hip_version", "nam_hip_model_load", "nam_hip_model_load_json", "nam_hip_model_load_ex",
    "nam_hip_model_free",
    "nam_hip_model_get_info", "nam_hip_model_slimmable_breakpoints", "nam_hip_batch_create", "nam_hip_batch_destroy",
    "nam_hip_batch_reset", "nam_hip_batch_set_slimmable_size", "nam_hip_batch_process_f32",
    "nam_hip_batch_process_f64", "nam_hip_batch_process_device", "nam_hip_batch_render_f32", "nam_hip_batch_synchronize",
    "nam_hip_batch_set_kernel", "nam_hip_batch_get_kernel", "nam_hip_batch_n_streams", "nam_hip_batch_kernel_name",
    "nam_hip_batch_set_persistent", "nam_hip_batch_flush", "nam_hip_batch_submit_f32", "nam_hip_batch_wait_f32", "nam_hip_batch_submit_f64", "nam_hip_batch_wait_f64",
    "nam_hip_batch_debug_timeline",
    "nam_hip_model_load_parts", "nam_hip_model_get_string", "nam_hip_model_get_weights", "nam_hip_sample_rate_from_nam",
    "nam_hip_batch_kernel_name_for", "nam_hip_version_support", "nam_hip_device_count",
]


class NamHipError(RuntimeError):
    """std::runtime_error of the reference's load path / any device failure."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[nam_hip {code}] {message}")
        self.code = code


class NamFileValidationError(NamHipError):
    """nam::NamFileValidationError (NAM/nam_file.h:11)."""


class _Info(ctypes.Structure):
    _fields_ = [
        ("architecture", ctypes.c_int32), ("in_channels", ctypes.c_int32), ("out_channels", ctypes.c_int32),
        ("prewarm_samples", ctypes.c_int32), ("expected_sample_rate", ctypes.c_double),
        ("has_loudness", ctypes.c_int32), ("has_input_level", ctypes.c_int32), ("has_output_level", ctypes.c_int32),
        ("is_slimmable", ctypes.c_int32), ("loudness", ctypes.c_double), ("input_level", ctypes.c_double),
        ("output_level", ctypes.c_double), ("num_weights", ctypes.c_int64), ("fast_tanh", ctypes.c_int32),
        ("has_a1_kernel", ctypes.c_int32), ("state_bytes_per_stream", ctypes.c_int64), ("version", ctypes.c_char * 32),
    ]


class _Lut(ctypes.Structure):
    _fields_ = [("function_name", ctypes.c_char_p), ("min_x", ctypes.c_float), ("max_x", ctypes.c_float),
                ("n_points", ctypes.c_int32)]


class _LoadOptions(ctypes.Structure):
    _fields_ = [("fast_tanh", ctypes.c_int32), ("n_luts", ctypes.c_int32), ("luts", ctypes.POINTER(_Lut)),
                ("version_checked_by_caller", ctypes.c_int32), ("struct_size", ctypes.c_int32)]


def lib_path() -> str:
    return os.path.join(_HERE, "lib", "libnam_hip.so")


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process. The PyTorch-ROCm wheel ships its own libamdhip64.so /
    libhsa-runtime64.so (same SONAME as /opt/rocm's). If libnam_hip.so pulled in /opt/rocm's copy
    first and torch then loaded its own, the second HSA runtime would find no GPU. So when torch is
    installed, make ITS runtime the resident one before libnam_hip.so is dlopen'ed; our DT_NEEDED
    libamdhip64.so.7 then binds to it. Stand-alone C/C++ callers simply get /opt/rocm's runtime.
    Set NAM_HIP_SYSTEM_RUNTIME=1 to skip this."""
    import sys
    if os.environ.get("NAM_HIP_SYSTEM_RUNTIME") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def load_library():
    """Load libnam_hip.so. Raises (loudly) if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"neuralampmodelercore_amd: {path} is missing — build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C neuralampmodelercore_amd/csrc`. "
            "There is no CPU fallback.")
    _share_torch_hip_runtime()
    L = ctypes.CDLL(path)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.nam_hip_last_error.restype = ctypes.c_char_p
    L.nam_hip_version.restype = ctypes.c_char_p
    L.nam_hip_model_load.argtypes = [ctypes.c_char_p, ci, ctypes.POINTER(vp)]
    L.nam_hip_model_load_json.argtypes = [ctypes.c_char_p, ci, ctypes.POINTER(vp)]
    L.nam_hip_model_load_ex.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(_LoadOptions), ctypes.POINTER(vp)]
    L.nam_hip_model_free.argtypes = [vp]
    L.nam_hip_model_free.restype = None
    L.nam_hip_model_get_info.argtypes = [vp, ctypes.POINTER(_Info)]
    L.nam_hip_model_slimmable_breakpoints.argtypes = [vp, ctypes.POINTER(cd), ci]
    L.nam_hip_batch_create.argtypes = [vp, ci, ci, ci, ctypes.POINTER(vp)]
    L.nam_hip_batch_destroy.argtypes = [vp]
    L.nam_hip_batch_destroy.restype = None
    L.nam_hip_batch_reset.argtypes = [vp, ci]
    L.nam_hip_batch_set_slimmable_size.argtypes = [vp, ctypes.POINTER(ci), ci, cd]
    L.nam_hip_batch_process_f32.argtypes = [vp, vp, vp, ci]
    L.nam_hip_batch_process_f64.argtypes = [vp, vp, vp, ci]
    L.nam_hip_batch_process_device.argtypes = [vp, vp, vp, ci, ctypes.c_int64, vp]
    L.nam_hip_batch_render_f32.argtypes = [vp, vp, vp, vp]
    L.nam_hip_batch_submit_f32.argtypes = [vp, vp, ci, ctypes.POINTER(ctypes.c_int64)]
    L.nam_hip_batch_wait_f32.argtypes = [vp, ctypes.c_int64, vp]
    L.nam_hip_batch_submit_f64.argtypes = [vp, vp, ci, ctypes.POINTER(ctypes.c_int64)]
    L.nam_hip_batch_wait_f64.argtypes = [vp, ctypes.c_int64, vp]
    L.nam_hip_batch_synchronize.argtypes = [vp]
    L.nam_hip_batch_set_kernel.argtypes = [vp, ci]
    L.nam_hip_batch_get_kernel.argtypes = [vp]
    L.nam_hip_batch_n_streams.argtypes = [vp]
    L.nam_hip_batch_set_persistent.argtypes = [vp, ci]
    L.nam_hip_batch_flush.argtypes = [vp, vp]
    L.nam_hip_batch_kernel_name.argtypes = [vp]
    L.nam_hip_batch_kernel_name.restype = ctypes.c_char_p
    L.nam_hip_batch_debug_timeline.argtypes = [vp, ci, vp]
    L.nam_hip_batch_kernel_name_for.argtypes = [vp, ci]
    L.nam_hip_batch_kernel_name_for.restype = ctypes.c_char_p
    L.nam_hip_model_load_parts.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, vp, ctypes.c_int64, cd,
                                           ctypes.POINTER(_LoadOptions), ctypes.POINTER(vp)]
    L.nam_hip_model_get_string.argtypes = [vp, ci, ctypes.c_char_p, ctypes.c_int64]
    L.nam_hip_model_get_string.restype = ctypes.c_int64
    L.nam_hip_model_get_weights.argtypes = [vp, vp, ctypes.c_int64]
    L.nam_hip_model_get_weights.restype = ctypes.c_int64
    L.nam_hip_version_support.argtypes = [ctypes.c_char_p]
    L.nam_hip_sample_rate_from_nam.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(cd)]
    _lib = L
    return L


def _check(rc: int):
    if rc >= 0:
        return rc
    msg = load_library().nam_hip_last_error().decode("utf-8", "replace")
    if rc == ERR_FILE:
        raise NamFileValidationError(rc, msg)
    raise NamHipError(rc, msg)


class Model:
    """A loaded .nam model (host side): what ``nam::get_dsp`` returns, minus the per-stream state."""

    def __init__(self, handle: int):
        self._L = load_library()
        self._h = ctypes.c_void_p(handle)
        info = _Info()
        _check(self._L.nam_hip_model_get_info(self._h, ctypes.byref(info)))
        self.info = info

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.nam_hip_model_free(self._h)
            self._h = None

    # --- nam::DSP getters (NAM/dsp.h:100-149) ---
    def NumInputChannels(self) -> int:
        return self.info.in_channels

    def NumOutputChannels(self) -> int:
        return self.info.out_channels

    def GetExpectedSampleRate(self) -> float:
        return self.info.expected_sample_rate

    def GetPrewarmSamples(self) -> int:
        return self.info.prewarm_samples

    def HasLoudness(self) -> bool:
        return bool(self.info.has_loudness)

    def GetLoudness(self) -> float:
        if not self.info.has_loudness:
            raise RuntimeError("Asked for loudness of a model that doesn't know how loud it is!")
        return self.info.loudness

    def HasInputLevel(self) -> bool:
        return bool(self.info.has_input_level)

    def GetInputLevel(self) -> float:
        return self.info.input_level

    def HasOutputLevel(self) -> bool:
        return bool(self.info.has_output_level)

    def GetOutputLevel(self) -> float:
        return self.info.output_level

    @property
    def architecture(self) -> str:
        return {1: "WaveNet", 2: "LSTM", 3: "SlimmableContainer"}.get(self.info.architecture, "?")

    @property
    def is_slimmable(self) -> bool:
        return bool(self.info.is_slimmable)

    @property
    def num_weights(self) -> int:
        return int(self.info.num_weights)

    @property
    def version(self) -> str:
        return self.info.version.decode()

    # --- nam::dspData (NAM/dsp.h:348-357) of the loaded model: what get_dsp(path, dspData& returnedConfig) hands back ---
    def _string(self, field: int) -> str:
        n = _check(self._L.nam_hip_model_get_string(self._h, field, None, 0))
        buf = ctypes.create_string_buffer(n + 1)
        _check(self._L.nam_hip_model_get_string(self._h, field, buf, n + 1))
        return buf.value.decode("utf-8", "replace")

    def dsp_data(self) -> dict:
        """{version, architecture, config (JSON text), metadata (JSON text), weights (float32 array), expected_sample_rate}."""
        n = _check(self._L.nam_hip_model_get_weights(self._h, None, 0))
        w = np.zeros(n, dtype=np.float32)
        if n:
            _check(self._L.nam_hip_model_get_weights(self._h, w.ctypes.data_as(ctypes.c_void_p), n))
        return dict(version=self._string(0), architecture=self._string(1), config=self._string(2), metadata=self._string(3),
                    weights=w, expected_sample_rate=self.info.expected_sample_rate)

    def describe(self) -> str:
        """One line about the device plans: which kernels take the model (and why nam_wn_reg_kernel does not)."""
        return self._string(4)

    def wr_why(self) -> str:
        """Why the register-resident WaveNet kernel refuses the full-size plan ('' when it takes it)."""
        import re
        m = re.findall(r"wn_reg=0 \(([^)]*)\)", self.describe())
        return m[-1] if m else ""

    def jit_failed(self) -> str:
        """Why the per-model compile of nam_wn_reg_kernel was not available ('' when it was, or was not needed)."""
        import re
        m = re.findall(r"wn_reg_jit=failed \((.*?)\)(?: \||$)", self.describe())
        return m[-1] if m else ""

    def GetSlimmableSizeBreakpoints(self) -> List[float]:
        buf = (ctypes.c_double * 64)()
        n = _check(self._L.nam_hip_model_slimmable_breakpoints(self._h, buf, 64))
        return [buf[i] for i in range(min(n, 64))]

    def batch(self, n_streams: int, max_frames: int = 64, device: int = 0) -> "Batch":
        return Batch(self, n_streams, max_frames, device)


class Batch:
    """N independent streams of one model on one GPU (replaces N ``nam::DSP`` instances)."""

    def __init__(self, model: Model, n_streams: int, max_frames: int, device: int = 0):
        self._L = load_library()
        self.model = model
        self.n_streams = int(n_streams)
        self.max_frames = int(max_frames)
        self.device = int(device)
        h = ctypes.c_void_p()
        _check(self._L.nam_hip_batch_create(model._h, device, n_streams, max_frames, ctypes.byref(h)))
        self._h = h
        self._ticket_frames = {}  # ticket -> n_frames of the buffers in flight (submit / wait)

    def close(self):
        if getattr(self, "_h", None):
            self._L.nam_hip_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    # DSP::Reset(sampleRate, maxBufferSize) — the buffer size was fixed at creation
    def Reset(self, prewarm: bool = True):
        _check(self._L.nam_hip_batch_reset(self._h, 1 if prewarm else 0))

    reset = Reset

    def SetSlimmableSize(self, ratio: float, stream_ids: Optional[Sequence[int]] = None):
        if stream_ids is None:
            _check(self._L.nam_hip_batch_set_slimmable_size(self._h, None, 0, float(ratio)))
        else:
            arr = (ctypes.c_int * len(stream_ids))(*[int(s) for s in stream_ids])
            _check(self._L.nam_hip_batch_set_slimmable_size(self._h, arr, len(stream_ids), float(ratio)))

    def set_kernel(self, kernel: int):
        _check(self._L.nam_hip_batch_set_kernel(self._h, int(kernel)))

    def get_kernel(self) -> int:
        return _check(self._L.nam_hip_batch_get_kernel(self._h))

    def set_persistent(self, enable: bool = True) -> bool:
        """Persistent block mode (include/nam_hip.h): True if the batch will use it."""
        return _check(self._L.nam_hip_batch_set_persistent(self._h, 1 if enable else 0)) == 1

    def flush(self, stream: int = 0):
        """Wait until every buffer submitted in persistent mode is rendered (``stream``: the hipStream_t they were issued on)."""
        _check(self._L.nam_hip_batch_flush(self._h, ctypes.c_void_p(stream)))

    def kernel_name(self, n_frames: Optional[int] = None) -> str:
        """The __global__ function the batch's largest stream group runs (the name rocprofv3 reports) for a launch of
        n_frames (default: one 64-frame buffer); under AUTO longer launches may run another kernel of the family."""
        if n_frames is None:
            return self._L.nam_hip_batch_kernel_name(self._h).decode()
        return self._L.nam_hip_batch_kernel_name_for(self._h, int(n_frames)).decode()

    def synchronize(self):
        _check(self._L.nam_hip_batch_synchronize(self._h))

    def debug_timeline(self, n_frames: int) -> np.ndarray:
        """Developer tool: [96, 8] int64; row w = wavefront w of workgroup 0 of the MFMA kernel's profiling build:
        barrier cycles, total cycles, then (compute waves) five per-job segment sums."""
        out = np.zeros((96, 8), dtype=np.int64)
        _check(self._L.nam_hip_batch_debug_timeline(self._h, int(n_frames), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def process(self, x: np.ndarray) -> np.ndarray:
        """x: host array [n_streams, in_channels, n_frames] (or [n_streams, n_frames] for mono),
        float32 or float64 (NAM_SAMPLE). Returns [n_streams, out_channels, n_frames], same dtype."""
        ic, oc = self.model.NumInputChannels(), self.model.NumOutputChannels()
        x = np.asarray(x)
        if x.ndim == 2:
            x = x[:, None, :]
        if x.shape[0] != self.n_streams or x.shape[1] != ic:
            raise ValueError(f"expected input [{self.n_streams}, {ic}, n], got {tuple(x.shape)}")
        n = x.shape[2]
        if x.dtype == np.float64:
            x = np.ascontiguousarray(x)
            out = np.empty((self.n_streams, oc, n), dtype=np.float64)
            fn = self._L.nam_hip_batch_process_f64
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            out = np.empty((self.n_streams, oc, n), dtype=np.float32)
            fn = self._L.nam_hip_batch_process_f32
        _check(fn(self._h, x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), n))
        return out

    PIPE_SLOTS = 16  # NAM_HIP_PIPE_SLOTS

    def submit(self, x: np.ndarray) -> int:
        """process() for callers that keep buffers in flight (nam_hip_batch_submit_f32): copies x ([n_streams, in_channels,
        n] or [n_streams, n], float32), starts the work, returns a ticket for wait(). Up to PIPE_SLOTS tickets at a time."""
        ic = self.model.NumInputChannels()
        x = np.asarray(x)
        if x.ndim == 2:
            x = x[:, None, :]
        if x.shape[0] != self.n_streams or x.shape[1] != ic:
            raise ValueError(f"expected input [{self.n_streams}, {ic}, n], got {tuple(x.shape)}")
        f64 = x.dtype == np.float64  # (NAM_SAMPLE = double callers: wait() hands float64 back)
        x = np.ascontiguousarray(x, dtype=np.float64 if f64 else np.float32)
        t = ctypes.c_int64(-1)
        fn = self._L.nam_hip_batch_submit_f64 if f64 else self._L.nam_hip_batch_submit_f32
        _check(fn(self._h, x.ctypes.data_as(ctypes.c_void_p), x.shape[2], ctypes.byref(t)))
        self._ticket_frames[t.value] = (x.shape[2], f64)
        return t.value

    def wait(self, ticket: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Blocks until the buffer behind `ticket` is rendered; returns [n_streams, out_channels, n] float32 (into `out` if given)."""
        entry = self._ticket_frames.get(ticket)
        if entry is None:  # (the library words the error)
            _check(self._L.nam_hip_batch_wait_f32(self._h, int(ticket), None))
            raise ValueError(f"ticket {ticket} is not in flight")
        n, f64 = entry
        dt = np.float64 if f64 else np.float32
        oc = self.model.NumOutputChannels()
        if out is None:
            out = np.empty((self.n_streams, oc, n), dtype=dt)
        elif out.dtype != dt or not out.flags.c_contiguous or out.size != self.n_streams * oc * n:
            raise ValueError("out: a C-contiguous array of n_streams x out_channels x n in the submit's sample type")
        fn = self._L.nam_hip_batch_wait_f64 if f64 else self._L.nam_hip_batch_wait_f32
        _check(fn(self._h, int(ticket), out.ctypes.data_as(ctypes.c_void_p)))
        del self._ticket_frames[ticket]
        return out

    def process_stream(self, x: np.ndarray, block: Optional[int] = None) -> np.ndarray:
        """Feed a long host signal in `block`-frame process() calls (what benchmodel / render do)."""
        block = block or self.max_frames
        x = np.asarray(x)
        if x.ndim == 2:
            x = x[:, None, :]
        outs = [self.process(x[:, :, s:s + block]) for s in range(0, x.shape[2], block)]
        return np.concatenate(outs, axis=2)

    def render(self, signals) -> list:
        """Offline re-amp of one whole (mono or [in_channels, n]) float32 signal per stream; lengths may differ.
        Returns a list of [out_channels, n_s] arrays (tools/render.cpp's block loop as one resident launch)."""
        ic, oc = self.model.NumInputChannels(), self.model.NumOutputChannels()
        if len(signals) != self.n_streams:
            raise ValueError(f"expected {self.n_streams} signals, got {len(signals)}")
        ins = [np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(ic, -1)) for x in signals]
        outs = [np.empty((oc, x.shape[1]), dtype=np.float32) for x in ins]
        P = ctypes.c_void_p * self.n_streams
        n = (ctypes.c_int64 * self.n_streams)(*[x.shape[1] for x in ins])
        _check(self._L.nam_hip_batch_render_f32(self._h, P(*[x.ctypes.data for x in ins]), P(*[y.ctypes.data for y in outs]), n))
        return outs

    def process_device(self, d_in: int, d_out: int, n_frames: int, frame_stride: Optional[int] = None,
                       stream: int = 0):
        """Raw device-pointer form (ints, e.g. torch.Tensor.data_ptr()). Enqueues only."""
        fs = n_frames if frame_stride is None else frame_stride
        _check(self._L.nam_hip_batch_process_device(self._h, ctypes.c_void_p(d_in), ctypes.c_void_p(d_out), n_frames,
                                                    fs, ctypes.c_void_p(stream) if stream else None))

    def process_tensor(self, x, out=None, n_frames: Optional[int] = None, stream=None):
        """torch.Tensor form: x [n_streams, in_ch, T] float32 on this batch's GPU; returns `out`
        [n_streams, out_ch, T]. Enqueued on `stream` (default: torch's current stream)."""
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
        T = x.shape[2]
        n = T if n_frames is None else n_frames
        if out is None:
            out = torch.empty((self.n_streams, self.model.NumOutputChannels(), T), dtype=torch.float32, device=x.device)
        assert out.is_contiguous() and out.shape[2] == T
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        self.process_device(x.data_ptr(), out.data_ptr(), n, T, s.cuda_stream)
        return out


def _load_ex(path, text, fast_tanh, luts) -> "Model":
    L = load_library()
    h = ctypes.c_void_p()
    items = list((luts or {}).items())
    arr = (_Lut * max(len(items), 1))()
    for i, (name, (lo, hi, n)) in enumerate(items):
        arr[i] = _Lut(name.encode(), float(lo), float(hi), int(n))
    opts = _LoadOptions(1 if fast_tanh else 0, len(items), arr, 0, ctypes.sizeof(_LoadOptions))
    _check(L.nam_hip_model_load_ex(os.fsencode(path) if path is not None else None,
                                   text.encode("utf-8") if text is not None else None, ctypes.byref(opts), ctypes.byref(h)))
    return Model(h.value)


def get_dsp(path: str, fast_tanh: bool = False, luts=None) -> Model:
    """nam::get_dsp(path). ``fast_tanh`` mirrors Activation::enable_fast_tanh() before loading; ``luts`` =
    {"Tanh" | "Sigmoid" | "SiLU": (min, max, n_points)} mirrors Activation::enable_lut(...) (activations.cpp:189-212)."""
    if luts:
        return _load_ex(path, None, fast_tanh, luts)
    L = load_library()
    h = ctypes.c_void_p()
    _check(L.nam_hip_model_load(os.fsencode(path), 1 if fast_tanh else 0, ctypes.byref(h)))
    return Model(h.value)


def get_dsp_json(text: str, fast_tanh: bool = False, luts=None) -> Model:
    """nam::get_dsp(json)."""
    if luts:
        return _load_ex(None, text, fast_tanh, luts)
    L = load_library()
    h = ctypes.c_void_p()
    _check(L.nam_hip_model_load_json(text.encode("utf-8"), 1 if fast_tanh else 0, ctypes.byref(h)))
    return Model(h.value)


def get_dsp_data(conf: dict, fast_tanh: bool = False) -> Model:
    """nam::get_dsp(dspData& conf) (NAM/get_dsp.h:91): `conf` as Model.dsp_data() returns it — version, architecture,
    config / metadata as JSON text, weights, expected_sample_rate."""
    L = load_library()
    h = ctypes.c_void_p()
    w = np.ascontiguousarray(conf["weights"], dtype=np.float32)
    opts = _LoadOptions(1 if fast_tanh else 0, 0, None, 0, ctypes.sizeof(_LoadOptions))
    md = conf.get("metadata")
    _check(L.nam_hip_model_load_parts(conf["version"].encode(), conf["architecture"].encode(), conf["config"].encode("utf-8"),
                                      md.encode("utf-8") if md else None, w.ctypes.data_as(ctypes.c_void_p), int(w.size),
                                      float(conf.get("expected_sample_rate", -1.0)), ctypes.byref(opts), ctypes.byref(h)))
    return Model(h.value)


def get_sample_rate_from_nam_file(path: Optional[str] = None, text: Optional[str] = None) -> float:
    """nam::get_sample_rate_from_nam_file (NAM/get_dsp.h:121): "sample_rate" of the document, or -1.0."""
    out = ctypes.c_double(0.0)
    _check(load_library().nam_hip_sample_rate_from_nam(os.fsencode(path) if path is not None else None,
                                                       text.encode("utf-8") if text is not None else None, ctypes.byref(out)))
    return out.value
