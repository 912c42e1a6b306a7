"""The drop-in boundary proven on the reference's OWN callers: /root/reference/tools/{benchmodel,
benchmodel_bufsize,loadmodel,render}.cpp compile UNMODIFIED against cpp/NAM/*.h (+ cpp/tools/wav.h for the AudioDSPTools
header render.cpp names) and link against libnam_hip.so (cpp/Makefile: ref_tools; the sources are compiled where they lie,
never copied). The CPU test does the compile here, where /root/reference exists; the binaries (cpp/ref_tools/, git-ignored)
travel to the GPU box with the snapshot, where the -m gpu tests run them on the reference's example models
(tools/benchmodel.cpp:9-11, 83-131; NAM/slimmable.h:13-29)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, model_path

REF = "/root/reference"
TOOLS = ("benchmodel", "benchmodel_bufsize", "loadmodel", "render")
OUT = os.path.join(ROOT, "cpp", "ref_tools")


def test_reference_tools_compile_unmodified_against_the_adapter(nam_lib):
    if not os.path.isdir(os.path.join(REF, "tools")):
        pytest.skip("the reference tree is not on this machine (the binaries were built where it is)")
    if "asan" in os.environ.get("LD_PRELOAD", ""):
        pytest.skip("sanitizer run of the host library (scripts/asan_host_check.sh): the tools would need the sanitizer runtime to link")
    r = subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "cpp"), "ref_tools"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    for t in TOOLS:
        # every compile line names the reference's own source file, not a copy
        assert re.search(rf"-o ref_tools/{t} {REF}/tools/{t}\.cpp ", r.stdout), r.stdout
        assert os.access(os.path.join(OUT, t), os.X_OK)
    # the headers the tools include beyond the standard library all resolve inside cpp/ (no reference header is on the path)
    assert "-I/root/reference" not in r.stdout and "-I" + REF not in r.stdout
    for t in TOOLS:
        src = open(os.path.join(REF, "tools", t + ".cpp")).read()
        for inc in re.findall(r'#include "([^"]+)"', src):
            assert os.path.exists(os.path.join(ROOT, "cpp", inc)) or os.path.exists(os.path.join(ROOT, "cpp", "tools", inc)), (t, inc)


def _tool(name):
    p = os.path.join(OUT, name)
    if not os.access(p, os.X_OK):
        pytest.skip("cpp/ref_tools was not built (needs /root/reference at build time)")
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["wavenet", "lstm", "slimmable_wavenet", "wavenet_a1_standard"])
def test_reference_benchmodel_runs_on_the_gpu(nam_lib, model):
    """the reference's benchmodel binary: load, Reset(expected rate, 64) incl. prewarm, 1,500 buffers of zeros through
    nam::DSP::process (tools/benchmodel.cpp:83-131)"""
    args = [_tool("benchmodel")]
    if model == "slimmable_wavenet":
        args += ["--slim", "0.5"]  # dynamic_cast<nam::SlimmableModel*> on the adapter's DSP (NAM/slimmable.h)
    r = subprocess.run(args + [model_path(model)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Fast tanh: enabled" in r.stdout and "Finished" in r.stdout, r.stdout
    ms = re.search(r"([0-9.]+)\s*ms", r.stdout)
    assert ms and float(ms.group(1)) > 0.0, r.stdout
    if model == "slimmable_wavenet":
        assert "Setting slimmable size to 0.5" in r.stdout
    # a model that is not slimmable refuses --slim the way the reference does
    if model == "lstm":
        bad = subprocess.run([_tool("benchmodel"), "--slim", "0.5", model_path(model)], capture_output=True, text=True, timeout=300)
        assert bad.returncode == 1 and "SlimmableModel" in bad.stderr


@pytest.mark.gpu
def test_reference_loadmodel_and_bufsize_run_on_the_gpu(nam_lib):
    for model in ("wavenet", "lstm", "slimmable_wavenet", "A2", "wavenet_a2_max"):
        r = subprocess.run([_tool("loadmodel"), model_path(model)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "Model loaded successfully" in r.stderr, r.stderr
    r = subprocess.run([_tool("benchmodel_bufsize"), model_path("wavenet"), "128", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_reference_render_matches_the_oracle(nam_lib, oracle, tmp_path):
    """the reference's render binary (tools/render.cpp:97-204) through the adapter: example_audio/input.wav through
    wavenet_a1_standard.nam, float32 WAV out, against the CPU oracle"""
    import struct
    wav = os.path.join(ROOT, "tests", "golden", "audio", "input.wav")
    out = str(tmp_path / "out.wav")
    r = subprocess.run([_tool("render"), model_path("wavenet_a1_standard"), wav, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    b = open(out, "rb").read()
    i = b.index(b"data")
    n = struct.unpack("<I", b[i + 4:i + 8])[0]
    y = np.frombuffer(b[i + 8:i + 8 + n], dtype="<f4")
    raw = open(wav, "rb").read()
    j = raw.index(b"data")
    m = struct.unpack("<I", raw[j + 4:j + 8])[0]
    q = np.frombuffer(raw[j + 8:j + 8 + m], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
    v = q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16)
    x = (np.where(v >= 1 << 23, v - (1 << 24), v) / 8388608.0).astype(np.float32)
    assert len(y) == len(x)
    ref = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=False)  # render leaves fast tanh off
    ref.Reset(48000.0, 64)
    want = ref.process_stream(x, 64)[0]
    assert float(np.max(np.abs(want - y))) <= 1e-4
