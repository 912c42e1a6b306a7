"""
nam_oracle.py — CPU ORACLE front end (test infrastructure, NOT product code).

Interprets a ``.nam`` file (JSON) the way the reference's loaders do and drives
the plain-C float32 numerics in ``nam_oracle.c`` through ctypes.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package never does.

Parity pinning: see the header of ``nam_oracle.c`` and DESIGN.md ("Oracle").

Reference lines followed (relative to /root/reference):
  * file validation            NAM/nam_file.cpp:9-40
  * version gate               NAM/get_dsp.cpp:18-39,113-128 ; get_dsp.h:66-67
  * sample rate / metadata     NAM/get_dsp.cpp:141-154,229-259,275-281
  * WaveNet config parsing     NAM/wavenet/model.cpp:913-1276
  * activation config parsing  NAM/activations.cpp:59-166
  * slimmable WaveNet          NAM/wavenet/slimmable.cpp:80-294,541-585
  * LSTM config                NAM/lstm.cpp:170-181
"""
from __future__ import annotations

import ctypes
import json
import math
import os
import re
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnam_oracle.so")

LATEST_FULLY_SUPPORTED_NAM_FILE_VERSION = "0.7.0"  # get_dsp.h:66
EARLIEST_SUPPORTED_NAM_FILE_VERSION = "0.5.0"  # get_dsp.h:67

ACT_TYPES = {
    "Identity": 0,
    "Tanh": 1,
    "Hardtanh": 2,
    "Fasttanh": 3,
    "ReLU": 4,
    "LeakyReLU": 5,
    "PReLU": 6,
    "Sigmoid": 7,
    "SiLU": 8,
    "Hardswish": 9,
    "LeakyHardtanh": 10,
    "LeakyHardTanh": 10,  # both casings accepted (activations.cpp:80)
    "Softsign": 11,
}
GATING = {"none": 0, "gated": 1, "blended": 2}
FILM_KEYS = [
    "conv_pre_film",
    "conv_post_film",
    "input_mixin_pre_film",
    "input_mixin_post_film",
    "activation_pre_film",
    "activation_post_film",
    "layer1x1_post_film",
    "head1x1_post_film",
]


class NamFileValidationError(Exception):
    """Mirror of nam::NamFileValidationError (NAM/nam_file.h:11)."""


def build(force: bool = False) -> str:
    """Compile nam_oracle.c -> libnam_oracle.so (gcc, no FMA contraction)."""
    src = os.path.join(_HERE, "nam_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        # -ffp-contract=off: every multiply/add individually rounded; -O3 vectorises across output
        # channels only (no -ffast-math), so results do not depend on the host's SIMD width.
        cmd = ["gcc", "-O3", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden",
               "-std=c99", "-o", _LIB_PATH, src, "-lm"]
        subprocess.check_call(cmd)
    return _LIB_PATH


def build_fast(out_path: str, opt: str = "-Ofast") -> str:
    """Timing-only build for bench.py's cpu_baseline: same source with the reference's Release flags
    (-Ofast, tools/CMakeLists.txt:176) — or `opt` = "-O2", BASELINE.md's primary figure — and -march=native. Built on the
    machine that runs it."""
    src = os.path.join(_HERE, "nam_oracle.c")
    subprocess.check_call(["gcc", opt, "-march=native", "-fPIC", "-shared", "-fvisibility=hidden", "-std=c99",
                           "-o", out_path, src, "-lm"])
    return out_path


_lib = None


def use_library(path: str) -> None:
    """Point the module at another build of the same C source (bench.py's -Ofast timing build)."""
    global _lib, _LIB_PATH
    _LIB_PATH = path
    _lib = None


def lib():
    global _lib
    if _lib is None:
        if _LIB_PATH == os.path.join(_HERE, "libnam_oracle.so"):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, ci, cf, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_long
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        L.orc_wavenet_new.restype = vp
        L.orc_wavenet_new.argtypes = [ci, ci]
        L.orc_wavenet_add_array.restype = ci
        L.orc_wavenet_add_array.argtypes = [vp] + [ci] * 15 + [ip, ci]
        L.orc_wavenet_add_layer.restype = ci
        L.orc_wavenet_add_layer.argtypes = [vp, ci, ci, ci, fp, ci, fp]
        L.orc_wavenet_set_head.restype = ci
        L.orc_wavenet_set_head.argtypes = [vp, ci, ci, ci, ip, ci, fp]
        L.orc_wavenet_set_condition_dsp.restype = None
        L.orc_wavenet_set_condition_dsp.argtypes = [vp, vp]
        L.orc_wavenet_expected_weights.restype = cl
        L.orc_wavenet_expected_weights.argtypes = [vp]
        L.orc_wavenet_finalize.restype = ci
        L.orc_wavenet_finalize.argtypes = [vp, fp, cl]
        for name in ("in_channels", "out_channels", "prewarm_samples"):
            f = getattr(L, "orc_wavenet_" + name)
            f.restype = ci
            f.argtypes = [vp]
        L.orc_wavenet_head_scale.restype = cf
        L.orc_wavenet_head_scale.argtypes = [vp]
        L.orc_wavenet_process.restype = None
        L.orc_wavenet_process.argtypes = [vp, fp, fp, ci]
        L.orc_wavenet_process_blocks.restype = None
        L.orc_wavenet_process_blocks.argtypes = [vp, fp, fp, cl, ci]
        L.orc_lstm_process_blocks.restype = None
        L.orc_lstm_process_blocks.argtypes = [vp, fp, fp, cl, ci]
        L.orc_wavenet_reset.restype = None
        L.orc_wavenet_reset.argtypes = [vp, ci, ci]
        L.orc_wavenet_free.restype = None
        L.orc_wavenet_free.argtypes = [vp]
        L.orc_lstm_expected_weights.restype = cl
        L.orc_lstm_expected_weights.argtypes = [ci, ci, ci, ci]
        L.orc_lstm_new.restype = vp
        L.orc_lstm_new.argtypes = [ci, ci, ci, ci, ci, fp, cl, ctypes.c_double, ci]
        L.orc_lstm_process.restype = None
        L.orc_lstm_process.argtypes = [vp, fp, fp, ci]
        L.orc_lstm_prewarm_samples.restype = ci
        L.orc_lstm_prewarm_samples.argtypes = [vp]
        L.orc_lstm_reset.restype = None
        L.orc_lstm_reset.argtypes = [vp, ci, ci]
        L.orc_lstm_free.restype = None
        L.orc_lstm_free.argtypes = [vp]
        L.orc_kat_conv1d.restype = ci
        L.orc_kat_conv1d.argtypes = [ci, ci, ci, ci, ci, ci, fp, cl, fp, fp, ci, ci, ci]
        L.orc_kat_conv1x1.restype = ci
        L.orc_kat_conv1x1.argtypes = [ci, ci, ci, ci, fp, cl, fp, fp, ci]
        L.orc_kat_film.restype = ci
        L.orc_kat_film.argtypes = [ci, ci, ci, ci, fp, cl, fp, fp, fp, ci]
        L.orc_kat_activation.restype = None
        L.orc_kat_activation.argtypes = [fp, fp, cl]
        L.orc_kat_gating.restype = None
        L.orc_kat_gating.argtypes = [ci, fp, fp, ci, fp, ci]
        L.orc_kat_layer.restype = ci
        L.orc_kat_layer.argtypes = [vp, fp, cl, fp, fp, fp, fp, ci]
        _lib = L
    return _lib


def _fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _iptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


# ---------------------------------------------------------------------------
# Version gate — get_dsp.cpp:18-39, 113-128
# ---------------------------------------------------------------------------
def _parse_version(v: str):
    a, b, c = v.split(".")
    return (int(a), int(b), int(c))


def version_support(version: str) -> str:
    """Returns 'yes' | 'partial' | 'no' (CoreVersionSupportChecker::support)."""
    if not re.match(r"^\d+\.\d+\.\d+$", version):
        return "no"
    parsed = _parse_version(version)
    latest = _parse_version(LATEST_FULLY_SUPPORTED_NAM_FILE_VERSION)
    earliest = _parse_version(EARLIEST_SUPPORTED_NAM_FILE_VERSION)
    if parsed < earliest:
        return "no"
    if parsed[0] > latest[0] or parsed[1] > latest[1]:
        return "no"
    if latest < parsed:
        return "partial"
    return "yes"


def verify_config_version(version: str) -> None:
    if version_support(version) == "no":
        raise RuntimeError(f"Model config is an unsupported version {version}.")


# ---------------------------------------------------------------------------
# Activation configs — activations.cpp:59-166
# ---------------------------------------------------------------------------
class LoadMode:
    """What the reference keeps in process-globals around get_dsp (activations.cpp:168-212): fast tanh and the
    lookup tables of Activation::enable_lut. Travels wherever a plain ``fast_tanh`` bool travels (truthiness = fast
    tanh); ``luts`` = {"Tanh" | "Sigmoid" | "SiLU": (min, max, n_points)}."""

    def __init__(self, fast_tanh: bool = False, luts=None):
        self.fast_tanh = bool(fast_tanh)
        self.luts = dict(luts or {})
        for name in self.luts:
            if name not in ("Tanh", "Sigmoid", "SiLU"):  # activations.cpp:209-211
                raise RuntimeError("Tried to enable LUT for a function other than Tanh, Sigmoid, or SiLU")

    def __bool__(self):
        return self.fast_tanh


def act_cfg(j, fast_tanh: bool = False) -> np.ndarray:
    """Encode an activation JSON (string or object) as the float cfg the C side takes:
    [type, p0, p1, p2, p3, n_slopes, slopes...]."""
    slopes: List[float] = []
    p = [0.0, 0.0, 0.0, 0.0]
    if j is None or j == "":
        t = 0
    elif isinstance(j, str):
        if j not in ACT_TYPES:
            raise RuntimeError("Unknown activation type: " + j)
        t = ACT_TYPES[j]
        if t == ACT_TYPES["LeakyReLU"]:
            p[0] = 0.01  # registry singleton _LEAKY_RELU(0.01)
        elif t == ACT_TYPES["PReLU"]:
            slopes = [0.01]
        elif t == ACT_TYPES["LeakyHardtanh"]:
            p = [-1.0, 1.0, 0.01, 0.01]
    elif isinstance(j, dict):
        name = j["type"]
        if name not in ACT_TYPES:
            raise RuntimeError("Unknown activation type: " + name)
        t = ACT_TYPES[name]
        if t == ACT_TYPES["PReLU"]:
            if "negative_slope" in j:
                slopes = [float(j["negative_slope"])]
            elif "negative_slopes" in j:
                slopes = [float(x) for x in j["negative_slopes"]]
            else:
                slopes = [0.01]
        elif t == ACT_TYPES["LeakyReLU"]:
            p[0] = float(j.get("negative_slope", 0.01))
        elif t == ACT_TYPES["LeakyHardtanh"]:
            p = [float(j.get("min_val", -1.0)), float(j.get("max_val", 1.0)),
                 float(j.get("min_slope", 0.01)), float(j.get("max_slope", 0.01))]
    else:
        raise RuntimeError("Invalid activation config: expected string or object")
    # Activation::enable_lut replaces the registry entry the model binds at construction (activations.cpp:189-212);
    # it wins over fast tanh (enable_fast_tanh() followed by enable_lut("Tanh", ...))
    for name, (lo, hi, n) in getattr(fast_tanh, "luts", {}).items():
        if t == ACT_TYPES[name]:
            out = np.zeros(7, dtype=np.float32)
            out[0] = 13  # ORC_ACT_LUT
            out[1:5] = [lo, hi, ACT_TYPES[name], n]
            return out
    # Activation::enable_fast_tanh swaps only the "Tanh" registry entry (activations.cpp:168-177)
    if fast_tanh and t == ACT_TYPES["Tanh"]:
        t = ACT_TYPES["Fasttanh"]
    out = np.zeros(6 + max(len(slopes), 1), dtype=np.float32)
    out[0] = t
    out[1:5] = p
    out[5] = len(slopes)
    out[6:6 + len(slopes)] = slopes
    return out


# ---------------------------------------------------------------------------
# WaveNet config parsing — model.cpp:913-1276
# ---------------------------------------------------------------------------
class LayerArrayParams:
    def __init__(self):
        self.input_size = 0
        self.condition_size = 0
        self.head_size = 0
        self.head_kernel_size = 1
        self.head_dilation = 1
        self.head_bias = False
        self.channels = 0
        self.bottleneck = 0
        self.kernel_sizes: List[int] = []
        self.dilations: List[int] = []
        self.activations: list = []
        self.gating_modes: List[int] = []
        self.secondary_activations: list = []
        self.groups_input = 1
        self.groups_input_mixin = 1
        self.layer1x1_active = True
        self.layer1x1_groups = 1
        self.head1x1_active = False
        self.head1x1_out = 0
        self.head1x1_groups = 1
        self.film = [[0, 0, 1] for _ in range(8)]  # active, shift, groups

    def copy(self) -> "LayerArrayParams":
        import copy
        return copy.deepcopy(self)


def parse_wavenet_layer(lc: dict, i: int) -> LayerArrayParams:
    p = LayerArrayParams()
    p.groups_input = int(lc.get("groups_input", 1))
    p.groups_input_mixin = int(lc.get("groups_input_mixin", 1))
    p.channels = int(lc["channels"])
    p.bottleneck = int(lc.get("bottleneck", p.channels))
    if "layer1x1" in lc:
        p.layer1x1_active = bool(lc["layer1x1"]["active"])
        p.layer1x1_groups = int(lc["layer1x1"]["groups"])
    p.input_size = int(lc["input_size"])
    p.condition_size = int(lc["condition_size"])
    # head: nested object preferred, legacy head_size/head_bias otherwise (model.cpp:961-990)
    if lc.get("head") is not None:
        hj = lc["head"]
        if not isinstance(hj, dict):
            raise RuntimeError(f"Layer array {i}: 'head' must be a JSON object")
        p.head_size = int(hj["out_channels"])
        if "head_dilation" in hj:
            p.head_dilation = int(hj["head_dilation"])
        p.head_kernel_size = int(hj["kernel_size"])
        p.head_bias = bool(hj["bias"])
    elif "head_size" in lc:
        p.head_size = int(lc["head_size"])
        p.head_kernel_size = 1
        p.head_bias = bool(lc["head_bias"])
    else:
        raise RuntimeError(f"Layer array {i}: expected 'head' object or legacy 'head_size' and 'head_bias'")
    if p.head_kernel_size < 1:
        raise RuntimeError(f"Layer array {i}: head.kernel_size must be >= 1")
    p.dilations = [int(d) for d in lc["dilations"]]
    n = len(p.dilations)
    has_ks, has_kss = "kernel_size" in lc, "kernel_sizes" in lc
    if has_ks and has_kss:
        raise RuntimeError(f"Layer array {i}: only one of kernel_size or kernel_sizes may be provided")
    if has_kss:
        if not isinstance(lc["kernel_sizes"], list):
            raise RuntimeError(f"Layer array {i}: kernel_sizes must be an array")
        p.kernel_sizes = [int(k) for k in lc["kernel_sizes"]]
        if len(p.kernel_sizes) != n:
            raise RuntimeError(f"Layer array {i}: kernel_sizes array size must match dilations size")
    elif has_ks:
        p.kernel_sizes = [int(lc["kernel_size"])] * n
    else:
        raise RuntimeError(f"Layer array {i}: either kernel_size or kernel_sizes must be provided")
    # activations (model.cpp:1038-1059)
    if isinstance(lc["activation"], list):
        p.activations = list(lc["activation"])
        if len(p.activations) != n:
            raise RuntimeError(f"Layer array {i}: activation array size must match dilations size")
    else:
        p.activations = [lc["activation"]] * n
    # gating (model.cpp:1061-1186)
    if "gating_mode" in lc:
        gm = lc["gating_mode"]
        if isinstance(gm, list):
            for idx, g in enumerate(gm):
                if g not in GATING:
                    raise RuntimeError("Invalid gating_mode: " + str(g))
                mode = GATING[g]
                p.gating_modes.append(mode)
                if mode != 0:
                    if "secondary_activation" in lc:
                        sa = lc["secondary_activation"]
                        if isinstance(sa, list):
                            if len(p.gating_modes) > len(sa):
                                raise RuntimeError(f"Layer array {i}: secondary_activation array too small")
                            p.secondary_activations.append(sa[len(p.gating_modes) - 1])
                        else:
                            p.secondary_activations.append(sa)
                    else:
                        p.secondary_activations.append("Sigmoid")
                else:
                    p.secondary_activations.append(None)
            if len(p.gating_modes) != n:
                raise RuntimeError(f"Layer array {i}: gating_mode array size must match dilations size")
            if isinstance(lc.get("secondary_activation"), list) and len(lc["secondary_activation"]) != n:
                raise RuntimeError(f"Layer array {i}: secondary_activation array size must match dilations size")
        else:
            if gm not in GATING:
                raise RuntimeError("Invalid gating_mode: " + str(gm))
            mode = GATING[gm]
            p.gating_modes = [mode] * n
            sa = None
            if mode != 0:
                sa = lc["secondary_activation"] if "secondary_activation" in lc else "Sigmoid"
            p.secondary_activations = [sa] * n
    elif "gated" in lc:
        gated = bool(lc["gated"])
        p.gating_modes = [1 if gated else 0] * n
        p.secondary_activations = ["Sigmoid" if gated else None] * n
    else:
        p.gating_modes = [0] * n
        p.secondary_activations = [None] * n
    # head1x1 (model.cpp:1188-1199)
    p.head1x1_out = p.channels
    if "head1x1" in lc:
        p.head1x1_active = bool(lc["head1x1"]["active"])
        p.head1x1_out = int(lc["head1x1"]["out_channels"])
        p.head1x1_groups = int(lc["head1x1"]["groups"])
    # FiLM (model.cpp:1201-1222)
    for k, key in enumerate(FILM_KEYS):
        if key not in lc or lc[key] is False:
            p.film[k] = [0, 0, 1]
        else:
            fc = lc[key]
            p.film[k] = [int(bool(fc.get("active", True))), int(bool(fc.get("shift", True))),
                         int(fc.get("groups", 1))]
    if p.film[6][0] and not p.layer1x1_active:
        raise RuntimeError(f"Layer array {i}: layer1x1_post_film cannot be active when layer1x1.active is false")
    # Layer ctor validations (detail.h:58-85)
    if not p.layer1x1_active and p.bottleneck != p.channels:
        raise ValueError("When layer1x1.active is false, bottleneck must equal channels")
    if p.film[7][0] and not p.head1x1_active:
        raise ValueError("Do not use post-head 1x1 FiLM if there is no head 1x1")
    return p


class WaveNetConfig:
    def __init__(self):
        self.arrays: List[LayerArrayParams] = []
        self.with_head = False
        self.head = None  # dict(in_channels, channels, out_channels, kernel_sizes, activation)
        self.head_scale = 0.0
        self.in_channels = 1
        self.condition_dsp_json = None


def parse_wavenet_config(config: dict) -> WaveNetConfig:
    wc = WaveNetConfig()
    if config.get("condition_dsp") is not None:
        wc.condition_dsp_json = config["condition_dsp"]
    for i, lc in enumerate(config["layers"]):
        wc.arrays.append(parse_wavenet_layer(lc, i))
    wc.with_head = config.get("head") is not None
    wc.head_scale = float(config["head_scale"])
    wc.in_channels = int(config.get("in_channels", 1))
    if not wc.arrays:
        raise RuntimeError("WaveNet config requires at least one layer array")
    if wc.with_head:
        hj = config["head"]
        implied_in = wc.arrays[-1].head_size
        if hj.get("in_channels") is not None and int(hj["in_channels"]) != implied_in:
            raise RuntimeError("WaveNet config: head.in_channels must equal last layer's head_size")
        ks = [int(k) for k in hj["kernel_sizes"]]
        if not ks:
            raise RuntimeError("WaveNet config: head.kernel_sizes must be non-empty")
        wc.head = dict(in_channels=implied_in, channels=int(hj["channels"]), out_channels=int(hj["out_channels"]),
                       kernel_sizes=ks, activation=hj["activation"])
    return wc


def _is_slimmable(config: dict) -> bool:
    """config_is_slimmable_wavenet — model.cpp:1290-1308"""
    layers = config.get("layers")
    if not isinstance(layers, list):
        return False
    for lc in layers:
        sl = lc.get("slimmable")
        if not isinstance(sl, dict):
            continue
        method = sl.get("method", "")
        if method != "slice_channels_uniform":
            if method:
                raise RuntimeError(f"SlimmableWavenet: unsupported slimmable method '{method}'")
            continue
        return True
    return False


# ---------------------------------------------------------------------------
# Slimmable weight extraction — slimmable.cpp:80-261
# ---------------------------------------------------------------------------
def ratio_to_channels(ratio: float, allowed: Sequence[int]) -> int:
    """slimmable.cpp:102-106"""
    idx = min(int(math.floor(ratio * float(len(allowed)))), len(allowed) - 1)
    return allowed[idx]


def _slim_bottleneck(p: LayerArrayParams, new_ch: int) -> int:
    """slimmable.cpp:80-85"""
    if not p.layer1x1_active:
        return new_ch
    return max(1, p.bottleneck * new_ch // p.channels)


def extract_slimmed_weights(arrays: List[LayerArrayParams], full: np.ndarray, new_channels: List[int]) -> np.ndarray:
    """slimmable.cpp:128-261 — keep the LEADING rows/cols of every channel-indexed dim."""
    pos = 0
    out: List[np.ndarray] = []

    def conv1x1(full_in, full_out, slim_in, slim_out, bias):
        nonlocal pos
        w = full[pos:pos + full_out * full_in].reshape(full_out, full_in)
        pos += full_out * full_in
        out.append(w[:slim_out, :slim_in].reshape(-1))
        if bias:
            b = full[pos:pos + full_out]
            pos += full_out
            out.append(b[:slim_out])

    def conv1d(full_in, full_out, slim_in, slim_out, k):
        nonlocal pos
        w = full[pos:pos + full_out * full_in * k].reshape(full_out, full_in, k)
        pos += full_out * full_in * k
        out.append(w[:slim_out, :slim_in, :].reshape(-1))
        b = full[pos:pos + full_out]
        pos += full_out
        out.append(b[:slim_out])

    def copy(n):
        nonlocal pos
        out.append(full[pos:pos + n])
        pos += n

    na = len(arrays)
    for arr, p in enumerate(arrays):
        if p.head_kernel_size != 1:
            raise RuntimeError("SlimmableWavenet: head rechannel kernel_size must be 1")
        if p.groups_input != 1 or p.groups_input_mixin != 1 or (p.layer1x1_active and p.layer1x1_groups != 1) or (
                p.head1x1_active and p.head1x1_groups != 1):
            raise RuntimeError("SlimmableWavenet: groups > 1 not supported")
        full_ch, full_bn = p.channels, p.bottleneck
        slim_ch = new_channels[arr]
        slim_bn = _slim_bottleneck(p, slim_ch)
        slim_input = p.input_size if arr == 0 else new_channels[arr - 1]
        slim_head = new_channels[arr + 1] if arr < na - 1 else p.head_size
        full_head_out = p.head1x1_out if p.head1x1_active else full_bn
        slim_head_out = p.head1x1_out if p.head1x1_active else slim_bn
        cs = p.condition_size
        conv1x1(p.input_size, full_ch, slim_input, slim_ch, False)
        for l in range(len(p.dilations)):
            gated = p.gating_modes[l] != 0
            full_bg = 2 * full_bn if gated else full_bn
            slim_bg = 2 * slim_bn if gated else slim_bn
            conv1d(full_ch, full_bg, slim_ch, slim_bg, p.kernel_sizes[l])
            conv1x1(cs, full_bg, cs, slim_bg, False)
            if p.layer1x1_active:
                conv1x1(full_bn, full_ch, slim_bn, slim_ch, True)
            if p.head1x1_active:
                conv1x1(full_bn, p.head1x1_out, slim_bn, p.head1x1_out, True)
            f = p.film
            m = lambda k: 2 if f[k][1] else 1
            if f[0][0]:
                conv1x1(cs, m(0) * full_ch, cs, m(0) * slim_ch, True)
            if f[1][0]:
                conv1x1(cs, m(1) * full_bg, cs, m(1) * slim_bg, True)
            if f[2][0]:
                dim = m(2) * cs
                copy(cs * dim + dim)
            if f[3][0]:
                conv1x1(cs, m(3) * full_bg, cs, m(3) * slim_bg, True)
            if f[4][0]:
                conv1x1(cs, m(4) * full_bg, cs, m(4) * slim_bg, True)
            if f[5][0]:
                conv1x1(cs, m(5) * full_bn, cs, m(5) * slim_bn, True)
            if f[6][0] and p.layer1x1_active:
                conv1x1(cs, m(6) * full_ch, cs, m(6) * slim_ch, True)
            if f[7][0] and p.head1x1_active:
                dim = m(7) * p.head1x1_out
                copy(cs * dim + dim)
        conv1x1(full_head_out, p.head_size, slim_head_out, slim_head, p.head_bias)
    copy(1)  # head_scale
    return np.ascontiguousarray(np.concatenate(out).astype(np.float32))


def modify_params_for_channels(arrays: List[LayerArrayParams], new_channels: List[int]) -> List[LayerArrayParams]:
    """slimmable.cpp:267-294"""
    out = []
    na = len(arrays)
    for i, p in enumerate(arrays):
        q = p.copy()
        q.channels = new_channels[i]
        q.bottleneck = _slim_bottleneck(p, new_channels[i])
        q.input_size = p.input_size if i == 0 else new_channels[i - 1]
        q.head_size = new_channels[i + 1] if i < na - 1 else p.head_size
        out.append(q)
    return out


# ---------------------------------------------------------------------------
# Model objects
# ---------------------------------------------------------------------------
class OracleDSP:
    """Common surface, shaped like nam::DSP (NAM/dsp.h:70-231) but planar float32 numpy I/O."""

    expected_sample_rate = -1.0
    loudness: Optional[float] = None
    input_level: Optional[float] = None
    output_level: Optional[float] = None
    max_buffer_size = 0

    def NumInputChannels(self) -> int:
        raise NotImplementedError

    def NumOutputChannels(self) -> int:
        raise NotImplementedError

    def GetPrewarmSamples(self) -> int:
        raise NotImplementedError

    def Reset(self, sample_rate: float, max_buffer_size: int, prewarm: bool = True) -> None:
        raise NotImplementedError

    def process(self, x: np.ndarray) -> np.ndarray:
        """x: [in_channels, n] (or [n] for mono) float32 -> [out_channels, n]; n <= max_buffer_size."""
        raise NotImplementedError

    def process_stream(self, x: np.ndarray, block: int) -> np.ndarray:
        """Run a long signal through in `block`-frame calls (what benchmodel/render do)."""
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float32)
        n = x.shape[1]
        out = np.zeros((self.NumOutputChannels(), n), dtype=np.float32)
        for s in range(0, n, block):
            e = min(n, s + block)
            out[:, s:e] = self.process(x[:, s:e])
        return out


class OracleWaveNet(OracleDSP):
    def __init__(self, wc: WaveNetConfig, weights: np.ndarray, sample_rate: float, fast_tanh: bool,
                 arrays: Optional[List[LayerArrayParams]] = None):
        L = lib()
        self._L = L
        self.expected_sample_rate = sample_rate
        self.fast_tanh = fast_tanh
        arrays = arrays if arrays is not None else wc.arrays
        self.arrays = arrays
        self._h = L.orc_wavenet_new(wc.in_channels, 1 if wc.with_head else 0)
        self._owned = True
        self._cond = None
        if wc.condition_dsp_json is not None:
            # nam::get_dsp(condition_dsp_json) — model.cpp:919-930
            cond = load_nam_json(wc.condition_dsp_json, fast_tanh=fast_tanh)
            if not isinstance(cond, OracleWaveNet):
                raise RuntimeError("oracle: condition_dsp must be a WaveNet")
            if cond.expected_sample_rate != sample_rate:
                raise RuntimeError("Condition DSP expected sample rate doesn't match WaveNet expected sample rate")
            if cond.NumInputChannels() != wc.in_channels:
                raise RuntimeError("input channels of WaveNet don't match input channels of condition DSP")
            self._cond = cond
            cond._owned = False  # freed by the parent's orc_wavenet_free
            L.orc_wavenet_set_condition_dsp(self._h, cond._h)
        for i, p in enumerate(arrays):
            if self._cond is not None and p.condition_size != self._cond.NumOutputChannels():
                raise RuntimeError(f"condition_size of layer {i} doesn't match output channels of condition DSP")
            if i > 0 and p.channels != arrays[i - 1].head_size:
                raise RuntimeError(f"channels of layer {i} doesn't match head_size of preceding layer")
            film = np.ascontiguousarray(np.array(p.film, dtype=np.int32).reshape(-1))
            idx = L.orc_wavenet_add_array(self._h, p.input_size, p.condition_size, p.head_size, p.head_kernel_size,
                                          p.head_dilation, int(p.head_bias), p.channels, p.bottleneck, p.groups_input,
                                          p.groups_input_mixin, int(p.layer1x1_active), p.layer1x1_groups,
                                          int(p.head1x1_active), p.head1x1_out, p.head1x1_groups, _iptr(film),
                                          len(p.dilations))
            assert idx == i
            for l in range(len(p.dilations)):
                a = act_cfg(p.activations[l], fast_tanh)
                a2 = act_cfg(p.secondary_activations[l], fast_tanh)
                L.orc_wavenet_add_layer(self._h, i, p.kernel_sizes[l], p.dilations[l], _fptr(a), p.gating_modes[l],
                                        _fptr(a2))
        if wc.with_head:
            h = wc.head
            ks = np.ascontiguousarray(np.array(h["kernel_sizes"], dtype=np.int32))
            a = act_cfg(h["activation"], fast_tanh)
            L.orc_wavenet_set_head(self._h, h["in_channels"], h["channels"], h["out_channels"], _iptr(ks), len(ks),
                                   _fptr(a))
        w = np.ascontiguousarray(weights, dtype=np.float32)
        self.expected_weights = int(L.orc_wavenet_expected_weights(self._h))
        if L.orc_wavenet_finalize(self._h, _fptr(w), len(w)) != 0:
            # model.cpp:671-682 (the "assigned" position is found by VALUE: first weight equal to the first unassigned one)
            if len(w) > self.expected_weights:
                pos = int(np.flatnonzero(w == w[self.expected_weights])[0])
                raise RuntimeError(f"Weight mismatch: assigned {pos + 1} weights, but {len(w)} were provided.")
            raise RuntimeError(f"Weight mismatch: provided {len(w)} weights, but the model expects more.")

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "_h", None):
            self._L.orc_wavenet_free(self._h)
            self._h = None

    def NumInputChannels(self):
        return self._L.orc_wavenet_in_channels(self._h)

    def NumOutputChannels(self):
        return self._L.orc_wavenet_out_channels(self._h)

    def GetPrewarmSamples(self):
        return self._L.orc_wavenet_prewarm_samples(self._h)

    @property
    def head_scale(self):
        return float(self._L.orc_wavenet_head_scale(self._h))

    def Reset(self, sample_rate, max_buffer_size, prewarm=True):
        self.max_buffer_size = max_buffer_size
        self._L.orc_wavenet_reset(self._h, max_buffer_size, 1 if prewarm else 0)

    def process(self, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float32)
        assert x.shape[0] == self.NumInputChannels()
        n = x.shape[1]
        assert n <= self.max_buffer_size, "num_frames must be <= max buffer size (model.cpp:824)"
        out = np.zeros((self.NumOutputChannels(), n), dtype=np.float32)
        self._L.orc_wavenet_process(self._h, _fptr(x), _fptr(out), n)
        return out

    def process_stream(self, x, block):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float32)
        assert x.shape[0] == self.NumInputChannels() and block <= self.max_buffer_size
        out = np.zeros((self.NumOutputChannels(), x.shape[1]), dtype=np.float32)
        self._L.orc_wavenet_process_blocks(self._h, _fptr(x), _fptr(out), x.shape[1], block)
        return out


class OracleSlimmableWaveNet(OracleDSP):
    """slimmable.cpp:352-530 — rebuilds a plain WaveNet of the selected width."""

    def __init__(self, config: dict, weights: np.ndarray, sample_rate: float, fast_tanh: bool):
        model_json = config["model"] if "model" in config else config
        self._wc = parse_wavenet_config(model_json)
        if self._wc.with_head:
            raise RuntimeError("SlimmableWavenet: post-stack head is not supported")
        self._full = np.ascontiguousarray(weights, dtype=np.float32)
        self.expected_sample_rate = sample_rate
        self.fast_tanh = fast_tanh
        self.allowed: List[List[int]] = []
        for lc in model_json["layers"]:
            al: List[int] = []
            sl = lc.get("slimmable")
            if isinstance(sl, dict):
                if sl.get("method", "") != "slice_channels_uniform":
                    raise RuntimeError("SlimmableWavenet: unsupported slimmable method")
                kw = sl.get("kwargs", {})
                if "allowed_channels" in kw:
                    al = [int(c) for c in kw["allowed_channels"]]
                else:
                    al = list(range(1, int(lc["channels"]) + 1))
            self.allowed.append(al)
        any_sl = False
        for i, al in enumerate(self.allowed):
            if al:
                any_sl = True
                if any(al[j] <= al[j - 1] for j in range(1, len(al))):
                    raise RuntimeError("SlimmableWavenet: allowed_channels must be sorted ascending")
                if al[-1] != self._wc.arrays[i].channels:
                    raise RuntimeError("SlimmableWavenet: last allowed_channels entry must equal the full channel count")
        if not any_sl:
            raise RuntimeError("SlimmableWavenet: at least one layer array must have allowed_channels")
        self._reset_args = None
        self._channels = None
        self._active = None
        self._rebuild([p.channels for p in self._wc.arrays])

    def channels_for(self, val: float) -> List[int]:
        return [p.channels if not al else ratio_to_channels(val, al) for p, al in zip(self._wc.arrays, self.allowed)]

    def slimmed_weights(self, channels: List[int]) -> np.ndarray:
        if channels == [p.channels for p in self._wc.arrays]:
            return self._full
        return extract_slimmed_weights(self._wc.arrays, self._full, channels)

    def _rebuild(self, channels: List[int]):
        if channels == self._channels and self._active is not None:
            return
        if channels == [p.channels for p in self._wc.arrays]:
            w, arrays = self._full, self._wc.arrays
        else:
            w = extract_slimmed_weights(self._wc.arrays, self._full, channels)
            arrays = modify_params_for_channels(self._wc.arrays, channels)
        self._active = OracleWaveNet(self._wc, w, self.expected_sample_rate, self.fast_tanh, arrays=arrays)
        self._channels = channels
        if self._reset_args is not None:
            self._active.Reset(*self._reset_args)

    def SetSlimmableSize(self, val: float):
        self._rebuild(self.channels_for(val))

    def GetSlimmableSizeBreakpoints(self) -> List[float]:
        bps = set()
        for al in self.allowed:
            for i in range(1, len(al)):
                bps.add(i / len(al))
        return sorted(bps)

    def NumInputChannels(self):
        return self._active.NumInputChannels()

    def NumOutputChannels(self):
        return self._active.NumOutputChannels()

    def GetPrewarmSamples(self):
        return 0  # slimmable.h:66 — the wrapper itself never prewarms; the active inner WaveNet does in its Reset

    def Reset(self, sample_rate, max_buffer_size, prewarm=True):
        self.max_buffer_size = max_buffer_size
        self._reset_args = (sample_rate, max_buffer_size, prewarm)
        self._active.Reset(sample_rate, max_buffer_size, prewarm)

    def process(self, x):
        return self._active.process(x)

    def process_stream(self, x, block):
        return self._active.process_stream(x, block)


class OracleLSTM(OracleDSP):
    def __init__(self, config: dict, weights: np.ndarray, sample_rate: float, fast_tanh: bool):
        L = lib()
        self._L = L
        self.expected_sample_rate = sample_rate
        self.num_layers = int(config["num_layers"])
        self.input_size = int(config["input_size"])
        self.hidden_size = int(config["hidden_size"])
        self.in_channels = int(config.get("in_channels", 1))
        self.out_channels = int(config.get("out_channels", 1))
        w = np.ascontiguousarray(weights, dtype=np.float32)
        self._h = L.orc_lstm_new(self.in_channels, self.out_channels, self.num_layers, self.input_size,
                                 self.hidden_size, _fptr(w), len(w), float(sample_rate), 1 if fast_tanh else 0)
        if not self._h:
            raise RuntimeError("LSTM weight count mismatch")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_lstm_free(self._h)
            self._h = None

    def NumInputChannels(self):
        return self.in_channels

    def NumOutputChannels(self):
        return self.out_channels

    def GetPrewarmSamples(self):
        return self._L.orc_lstm_prewarm_samples(self._h)

    def Reset(self, sample_rate, max_buffer_size, prewarm=True):
        self.max_buffer_size = max_buffer_size
        self._L.orc_lstm_reset(self._h, max_buffer_size, 1 if prewarm else 0)

    def process(self, x):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float32)
        n = x.shape[1]
        out = np.zeros((self.out_channels, n), dtype=np.float32)
        self._L.orc_lstm_process(self._h, _fptr(x), _fptr(out), n)
        return out

    def process_stream(self, x, block):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float32)
        out = np.zeros((self.out_channels, x.shape[1]), dtype=np.float32)
        self._L.orc_lstm_process_blocks(self._h, _fptr(x), _fptr(out), x.shape[1], block)
        return out


class OracleContainer(OracleDSP):
    """SlimmableContainer — NAM/container.cpp:17-144. Submodels are whole .nam documents; one is active at a time
    (the last one on construction), chosen by SetSlimmableSize: first max_value with val < max_value, else last."""

    def __init__(self, config: dict, sample_rate: float, fast_tanh: bool):
        subs = config.get("submodels")
        if not isinstance(subs, list) or not subs:
            raise RuntimeError("SlimmableContainer: 'submodels' must be a non-empty array")
        self.expected_sample_rate = sample_rate
        self._max = [float(e["max_value"]) for e in subs]
        self._models = [load_nam_json(e["model"], fast_tanh=fast_tanh) for e in subs]
        for i in range(1, len(self._max)):  # container.cpp:26-30
            if self._max[i] <= self._max[i - 1]:
                raise RuntimeError("ContainerModel: submodels must be sorted by ascending max_value")
        if self._max[-1] < 1.0:  # :31-32
            raise RuntimeError("ContainerModel: last submodel max_value must be >= 1.0")
        for m in self._models:  # :35-46
            sr = m.expected_sample_rate
            if sr != sample_rate and sr != -1.0 and sample_rate != -1.0:
                raise RuntimeError(f"ContainerModel: submodel sample rate mismatch (expected {sample_rate:g}, got {sr:g})")
        self._active = len(self._models) - 1  # :49
        self._reset_args = None

    def index_for(self, val: float) -> int:  # container.cpp:103-115
        for i, mx in enumerate(self._max):
            if val < mx:
                return i
        return len(self._max) - 1

    def SetSlimmableSize(self, val: float):  # :117-139: a change of submodel Resets the newly active one
        i = self.index_for(val)
        if i == self._active:
            return
        if self._reset_args is not None:
            self._models[i].Reset(*self._reset_args)
        self._active = i

    def GetSlimmableSizeBreakpoints(self) -> List[float]:
        return list(self._max[:-1])

    def NumInputChannels(self):
        return 1  # container.cpp:19

    def NumOutputChannels(self):
        return 1

    def GetPrewarmSamples(self):
        return self._models[self._active].GetPrewarmSamples()

    def Reset(self, sample_rate, max_buffer_size, prewarm=True):  # :85-101: only the active submodel
        self.max_buffer_size = max_buffer_size
        self._reset_args = (sample_rate, max_buffer_size, prewarm)
        self._models[self._active].Reset(sample_rate, max_buffer_size, prewarm)

    def process(self, x):
        return self._models[self._active].process(x)

    def process_stream(self, x, block):
        return self._models[self._active].process_stream(x, block)


# ---------------------------------------------------------------------------
# get_dsp — get_dsp.cpp:141-273
# ---------------------------------------------------------------------------
def load_nam_json(j: dict, fast_tanh: bool = False) -> OracleDSP:
    verify_config_version(j["version"])
    if "weights" not in j:
        raise RuntimeError("Corrupted model file is missing weights.")
    weights = np.asarray(j["weights"], dtype=np.float32)
    arch = j["architecture"]
    config = j["config"]
    sample_rate = float(j["sample_rate"]) if "sample_rate" in j else -1.0
    if arch == "WaveNet":
        # dispatch order: slimmable -> (a2_fast: same math as generic) -> generic — model.cpp:1312-1326
        if _is_slimmable(config):
            dsp = OracleSlimmableWaveNet(config, weights, sample_rate, fast_tanh)
        else:
            dsp = OracleWaveNet(parse_wavenet_config(config), weights, sample_rate, fast_tanh)
    elif arch == "LSTM":
        dsp = OracleLSTM(config, weights, sample_rate, fast_tanh)
    elif arch == "SlimmableContainer":
        dsp = OracleContainer(config, sample_rate, fast_tanh)
    else:
        raise RuntimeError("No config parser registered for architecture: " + str(arch))
    md = j.get("metadata")
    if md:
        for key, attr in (("loudness", "loudness"), ("input_level_dbu", "input_level"),
                          ("output_level_dbu", "output_level")):
            if md.get(key) is not None:
                setattr(dsp, attr, float(md[key]))
    return dsp


def validate_nam_file(path: str) -> dict:
    """nam_file.cpp:9-40"""
    if not os.path.exists(path):
        raise NamFileValidationError(f"Could not validate .nam file [{path}]: file does not exist.")
    try:
        with open(path, "r") as f:
            j = json.load(f)
    except OSError:
        raise NamFileValidationError(f"Could not validate .nam file [{path}]: file could not be read.")
    except json.JSONDecodeError as e:
        raise NamFileValidationError(f"Could not parse .nam file [{path}]: {e}")
    if not isinstance(j, dict):
        raise NamFileValidationError(f"Invalid .nam file [{path}]: root JSON value must be an object.")
    for key in ("version", "architecture", "config", "weights"):
        if key not in j:
            raise NamFileValidationError(f'Invalid .nam file [{path}]: missing required key "{key}".')
    return j


def get_dsp(path: str, fast_tanh: bool = False, luts=None) -> OracleDSP:
    """``luts``: see LoadMode (Activation::enable_lut before get_dsp)."""
    return load_nam_json(validate_nam_file(path), fast_tanh=LoadMode(fast_tanh, luts) if luts else fast_tanh)
