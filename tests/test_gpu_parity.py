"""-m gpu parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs. Tolerances follow the reference's own alt-implementation precedent
(tools/test/test_a2_fast.cpp:296-298: max-abs 5e-5) — 5e-5 with fast_tanh (a pure rational
function), 1e-4 with libm tanh / expf (device vs host transcendental ulp differences)."""
import os

import numpy as np
import pytest

from conftest import model_path
from signals import stream_bank, two_tone

pytestmark = pytest.mark.gpu

WAVENETS = ["wavenet", "wavenet_a1_standard", "wavenet_a2_max", "wavenet_condition_dsp", "slimmable_wavenet", "synth_a1_nano"]
# synthetic A1-family fixtures (tests/golden/make_synthetic_models.py) that reach the MFMA kernel's variants
SYNTH_A1 = ["synth_a1_13", "synth_a1_c12", "synth_a1_c8", "synth_a1_mixed", "synth_a1_lite", "synth_a1_c14", "synth_a1_feather"]
# single-array fixtures with per-layer kernel sizes 1..16 and a head rechannel with taps: the K-tap MFMA kernel
SYNTH_KT = ["synth_kt_c8", "synth_kt_c16", "synth_kt_c12", "synth_kt_c4"]


def _oracle_run(oracle, name, x, block, fast_tanh, ratio=None):
    ref = oracle.get_dsp(model_path(name), fast_tanh=fast_tanh)
    if ratio is not None:
        ref.SetSlimmableSize(ratio)
    ref.Reset(48000.0, block)
    return ref.process_stream(x, block)


def _tol(fast_tanh):
    return 5e-5 if fast_tanh else 1e-4


@pytest.mark.parametrize("name", WAVENETS)
@pytest.mark.parametrize("fast_tanh", [True, False])
@pytest.mark.parametrize("kernel", ["generic", "a1", "auto", "wn_reg"])
def test_wavenet_matches_oracle(nam_lib, oracle, name, fast_tanh, kernel):
    nam = nam_lib
    n_streams, block, n = 5, 64, 64 * 6
    x = stream_bank(n_streams, n, seed=3)
    model = nam.get_dsp(model_path(name), fast_tanh=fast_tanh)
    batch = model.batch(n_streams, block)
    if kernel == "generic":
        batch.set_kernel(nam.KERNEL_GENERIC)
    elif kernel == "a1":
        if not (model.info.has_a1_kernel & 1):
            pytest.skip("model is outside the A1 family")
        batch.set_kernel(nam.KERNEL_A1)
    elif kernel == "wn_reg":
        if not (model.info.has_a1_kernel & 16):
            pytest.skip("model is outside the register-resident kernel's shapes")
        batch.set_kernel(nam.KERNEL_WN_REG)
        assert batch.kernel_name() == "nam_wn_reg_kernel"
    batch.Reset(prewarm=True)
    y = batch.process_stream(x, block)
    assert y.shape == (n_streams, model.NumOutputChannels(), n)
    for s in range(n_streams):
        r = _oracle_run(oracle, name, x[s], block, fast_tanh)
        scale = max(1.0, float(np.max(np.abs(r))))
        err = float(np.max(np.abs(r - y[s])))
        assert err <= _tol(fast_tanh) * scale, (name, s, err, scale)
    batch.close()


@pytest.mark.parametrize("name", SYNTH_A1)
@pytest.mark.parametrize("fast_tanh", [True, False])
def test_mfma_kernel_variants_match_oracle(nam_lib, oracle, name, fast_tanh):
    """Idle job + padded launch (13 layers), prefetch depth 5 and 6, 12- / 4-channel full layout, half layout
    end to end, run-time activation dispatch — block launches, a ragged tail and one multi-block launch."""
    nam = nam_lib
    n_streams, block, n = 3, 64, 64 * 5 + 17
    x = stream_bank(n_streams, n, seed=31)
    model = nam.get_dsp(model_path(name), fast_tanh=fast_tanh)
    assert model.info.has_a1_kernel & 2, "fixture must be MFMA-eligible"
    for mode, max_frames in (("blocks", block), ("one_launch", 512)):
        refs = [_oracle_run(oracle, name, x[s], max_frames, fast_tanh) for s in range(n_streams)]
        batch = model.batch(n_streams, max_frames)
        batch.set_kernel(nam.KERNEL_A1_MFMA)
        assert batch.get_kernel() == nam.KERNEL_A1_MFMA
        batch.Reset(prewarm=True)
        y = batch.process_stream(x, max_frames)
        for s in range(n_streams):
            scale = max(1.0, float(np.max(np.abs(refs[s]))))
            err = float(np.max(np.abs(refs[s] - y[s])))
            assert err <= _tol(fast_tanh) * scale, (name, mode, s, err, scale)
        batch.close()


@pytest.mark.parametrize("name", SYNTH_KT + ["A2"])
@pytest.mark.parametrize("fast_tanh", [True, False])
def test_ktap_mfma_kernel_matches_oracle(nam_lib, oracle, name, fast_tanh):
    """nam_kt_mfma_kernel (A2 shapes): half / full lane layout, 1-3 chunks per layer, lookbacks inside, across and far
    beyond a block, head rechannel with 1 / 3 / 5 / 16 taps (one at dilation 2), compile-time and run-time activation
    dispatch — block launches with a ragged tail and multi-block launches; the VALU kernel on the same state layout
    must agree, and the two must be interchangeable mid-stream."""
    nam = nam_lib
    n_streams, block, n = 3, 64, 64 * 7 + 23
    x = stream_bank(n_streams, n, seed=37)
    model = nam.get_dsp(model_path(name), fast_tanh=fast_tanh)
    assert model.info.has_a1_kernel & 2, "fixture must be MFMA-eligible"
    for mode, max_frames in (("blocks", block), ("one_launch", 512)):
        refs = [_oracle_run(oracle, name, x[s], max_frames, fast_tanh) for s in range(n_streams)]
        ys = {}
        for kernel in (nam.KERNEL_A1_MFMA, nam.KERNEL_A1):
            batch = model.batch(n_streams, max_frames)
            batch.set_kernel(kernel)
            assert batch.get_kernel() == kernel
            batch.Reset(prewarm=True)
            ys[kernel] = batch.process_stream(x, max_frames)
            batch.close()
        for kernel, y in ys.items():
            for s in range(n_streams):
                scale = max(1.0, float(np.max(np.abs(refs[s]))))
                err = float(np.max(np.abs(refs[s] - y[s])))
                assert err <= _tol(fast_tanh) * scale, (name, mode, kernel, s, err, scale)
    # switch kernels between blocks of one stream: same rings, same write positions
    batch = model.batch(n_streams, block)
    batch.Reset(prewarm=True)
    parts = []
    for i, k0 in enumerate(range(0, 64 * 6, 64)):
        batch.set_kernel(nam.KERNEL_A1_MFMA if i % 2 == 0 else nam.KERNEL_A1)
        parts.append(batch.process(x[:, k0:k0 + 64]))
    y = np.concatenate(parts, axis=-1)
    batch.close()
    refs = [_oracle_run(oracle, name, x[s, :64 * 6], block, fast_tanh) for s in range(n_streams)]
    for s in range(n_streams):
        assert float(np.max(np.abs(refs[s] - y[s]))) <= _tol(fast_tanh) * max(1.0, float(np.max(np.abs(refs[s]))))


def test_fuzzed_models_all_kernels(nam_lib, oracle):
    """tools/fuzz_models.py: 24 seeded random A1-family models (random kernel sizes 1..16, dilations up to 700, head
    taps, channel counts 1 .. 16, activations, array counts) through every kernel that accepts them — the register-resident
    kernel's LDS rings included —, block launches and one multi-block launch, against the oracle."""
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_models.py"), "24", "23"], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "FUZZ OK" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_lstm_matches_oracle(nam_lib, oracle):
    nam = nam_lib
    for fast_tanh in (True, False):
        n_streams, block, n = 70, 64, 64 * 4  # 70: exercises a partial 64-stream wavefront
        rng = np.random.default_rng(0)
        x = rng.uniform(-0.5, 0.5, size=(n_streams, n)).astype(np.float32)
        model = nam.get_dsp(model_path("lstm"), fast_tanh=fast_tanh)
        assert model.GetPrewarmSamples() == 24000
        batch = model.batch(n_streams, block)
        batch.Reset(prewarm=True)
        y = batch.process_stream(x, block)
        for s in (0, 1, 63, 64, 69):
            r = _oracle_run(oracle, "lstm", x[s], block, fast_tanh)
            err = float(np.max(np.abs(r - y[s])))
            assert err <= _tol(fast_tanh), (s, err)
        batch.close()


def test_block_partition_independence(nam_lib):
    """process(n) in one call == the same audio in ragged calls (state carried in HBM rings)."""
    nam = nam_lib
    x = stream_bank(3, 500, seed=5)
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    for kernel in (nam.KERNEL_GENERIC, nam.KERNEL_A1, nam.KERNEL_A1_MFMA):
        b1 = model.batch(3, 512)
        b1.set_kernel(kernel)
        b1.Reset(prewarm=False)
        y1 = b1.process(x)
        b2 = model.batch(3, 512)
        b2.set_kernel(kernel)
        b2.Reset(prewarm=False)
        parts, pos = [], 0
        for n in (1, 63, 64, 65, 7, 200, 100):
            parts.append(b2.process(x[:, pos:pos + n]))
            pos += n
        y2 = np.concatenate(parts, axis=2)
        assert pos == 500
        np.testing.assert_array_equal(y1, y2)
        b1.close()
        b2.close()


def test_generic_and_a1_kernels_agree(nam_lib):
    nam = nam_lib
    x = stream_bank(4, 64 * 5, seed=9)
    for name in ("wavenet", "wavenet_a1_standard", "slimmable_wavenet", "A2"):  # A2: K = 6 / 15 taps, head K = 16
        model = nam.get_dsp(model_path(name), fast_tanh=True)
        assert model.info.has_a1_kernel & 1
        ys = []
        kernels = [nam.KERNEL_GENERIC, nam.KERNEL_A1] + ([nam.KERNEL_A1_MFMA] if model.info.has_a1_kernel & 2 else [])
        for kernel in kernels:
            b = model.batch(4, 64)
            b.set_kernel(kernel)
            b.Reset(prewarm=True)
            ys.append(b.process_stream(x, 64))
            b.close()
        for y in ys[1:]:
            assert float(np.max(np.abs(ys[0] - y))) < 1e-5 * max(1.0, float(np.max(np.abs(ys[0]))))


def test_double_api_matches_float(nam_lib):
    nam = nam_lib
    x = stream_bank(2, 128, seed=1)
    model = nam.get_dsp(model_path("wavenet"), fast_tanh=False)
    b = model.batch(2, 64)
    b.Reset()
    yf = b.process_stream(x, 64)
    b.Reset()
    yd = b.process_stream(x.astype(np.float64), 64)
    assert yd.dtype == np.float64
    np.testing.assert_array_equal(yf.astype(np.float64), yd)
    b.close()


def test_too_many_frames_is_an_error(nam_lib):
    nam = nam_lib
    model = nam.get_dsp(model_path("wavenet"))
    b = model.batch(1, 64)
    with pytest.raises(nam.NamHipError):
        b.process(np.zeros((1, 1, 65), dtype=np.float32))
    b.close()


def test_slimmable_mixed_width_batch(nam_lib, oracle):
    """Config 5: streams of one batch run at different widths; each matches the oracle's slimmed model."""
    nam = nam_lib
    n_streams, block, n = 8, 64, 64 * 5
    x = stream_bank(n_streams, n, seed=11)
    ratios = [0.0, 0.34, 0.67, 1.0]
    model = nam.get_dsp(model_path("slimmable_wavenet"), fast_tanh=False)
    assert model.is_slimmable and model.GetSlimmableSizeBreakpoints() == pytest.approx([1 / 3, 2 / 3])
    b = model.batch(n_streams, block)
    b.Reset(prewarm=True)
    for i, r in enumerate(ratios):
        b.SetSlimmableSize(r, [s for s in range(n_streams) if s % len(ratios) == i])
    y = b.process_stream(x, block)
    for s in range(n_streams):
        r = _oracle_run(oracle, "slimmable_wavenet", x[s], block, False, ratio=ratios[s % len(ratios)])
        scale = max(1.0, float(np.max(np.abs(r))))
        assert float(np.max(np.abs(r - y[s]))) <= 1e-4 * scale, s
    b.close()


def test_device_pointer_long_render(nam_lib, oracle):
    """Offline re-amp form: whole signal resident in HBM, one launch walks it in 64-frame blocks."""
    torch = pytest.importorskip("torch")
    nam = nam_lib
    n_streams, n = 6, 64 * 20 + 17
    x = stream_bank(n_streams, n, seed=4)
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    b = model.batch(n_streams, 64)
    b.Reset(prewarm=True)
    xd = torch.from_numpy(x[:, None, :]).cuda()
    yd = b.process_tensor(xd)
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    for s in (0, n_streams - 1):
        r = _oracle_run(oracle, "wavenet_a1_standard", x[s], 64, True)
        assert float(np.max(np.abs(r - y[s]))) <= 5e-5
    b.close()


def test_hip_matches_committed_golden_vectors(nam_lib):
    """HIP path vs the committed fixtures (tests/golden/outputs.npz), all kernels that can run each model."""
    import os
    from conftest import ROOT
    nam = nam_lib
    G = np.load(os.path.join(ROOT, "tests", "golden", "outputs.npz"))
    x = G["input"]
    for key in [k for k in G.files if "__ft" in k]:
        name, ft = key.split("__ft")
        model = nam.get_dsp(model_path(name), fast_tanh=bool(int(ft)))
        kernels = [nam.KERNEL_GENERIC]
        if model.architecture == "WaveNet" and model.info.has_a1_kernel & 1:
            kernels.append(nam.KERNEL_A1)
        if model.architecture == "WaveNet" and model.info.has_a1_kernel & 2:
            kernels.append(nam.KERNEL_A1_MFMA)
        for kernel in kernels:
            b = model.batch(3, 64)
            if model.architecture == "WaveNet":
                b.set_kernel(kernel)
            b.Reset(prewarm=True)
            y = b.process_stream(np.stack([x, x, x]), 64)
            scale = max(1.0, float(np.max(np.abs(G[key]))))
            tol = (5e-5 if int(ft) else 1e-4) * scale
            for s in range(3):
                assert float(np.max(np.abs(y[s] - G[key]))) <= tol, (key, kernel, s)
            b.close()


@pytest.mark.parametrize("kernel", ["generic", "a1", "a1_mfma"])
def test_long_resident_render_wraps_every_ring(nam_lib, oracle, kernel):
    """One launch over 150 blocks: the d = 512 ring (1088 frames) wraps ~9 times; frames written by one
    wavefront are read back by others many blocks later (same-CU L1 coherence of global memory)."""
    torch = pytest.importorskip("torch")
    nam = nam_lib
    n_streams, n = 3, 64 * 150 + 5
    x = stream_bank(n_streams, n, seed=21)
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    b = model.batch(n_streams, 64)
    b.set_kernel({"generic": nam.KERNEL_GENERIC, "a1": nam.KERNEL_A1, "a1_mfma": nam.KERNEL_A1_MFMA}[kernel])
    b.Reset(prewarm=True)
    xd = torch.from_numpy(x[:, None, :]).cuda()
    y = b.process_tensor(xd)
    torch.cuda.synchronize()
    y = y.cpu().numpy()
    r = _oracle_run(oracle, "wavenet_a1_standard", x[1], 64, True)
    assert float(np.max(np.abs(r - y[1]))) <= 5e-5
    # and the state left behind continues correctly in block mode
    x2 = stream_bank(n_streams, 128, seed=22)
    y2 = b.process_stream(x2, 64)
    ref = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    ref.Reset(48000.0, 64)
    ref.process_stream(x[1], 64)
    r2 = ref.process_stream(x2[1], 64)
    assert float(np.max(np.abs(r2 - y2[1]))) <= 5e-5
    b.close()


def test_cpp_adapter_benchmodel_runs(nam_lib):
    """The C++ adapter (cpp/NAM/dsp.h: nam::get_dsp / nam::DSP::process over the C ABI) as a stand-alone
    binary, the way tools/benchmodel.cpp drives the reference."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "cpp", "tools", "benchmodel")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cpp")])
    for args in ([model_path("wavenet")], [model_path("lstm")], [model_path("slimmable_wavenet"), "--slim", "0.5"],
                 [model_path("wavenet_a1_standard"), "--streams", "64"], [model_path("A2")], [model_path("A2"), "--slim", "0.2"],
                 # device-resident buffers through BatchDSP::process_device / flush (sessions; 300 streams: more than CUs)
                 [model_path("wavenet_a1_standard"), "--streams", "300", "--resident"], [model_path("wavenet_a2_max"), "--streams", "40", "--resident"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert "ms" in out.stdout
    bad = subprocess.run([exe, "/nonexistent.nam"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "does not exist" in bad.stderr
    # real-time safety (tools/test/allocation_tracking.cpp:21-90 asserts zero allocations inside process()): neither the
    # adapter nor libnam_hip.so asks for heap memory in the steady-state loop, one stream and many; the per-buffer round
    # trip is printed
    import re
    for args in ([model_path("wavenet_a1_standard"), "--count-allocs"], [model_path("lstm"), "--count-allocs"],
                 [model_path("wavenet_a2_max"), "--count-allocs", "--streams", "16"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert "round trip per buffer (us): min" in out.stdout
        m = re.search(r"nam_hip\+adapter (\d+)", out.stdout)
        assert m and int(m.group(1)) == 0, out.stdout


def _read_f32_wav(path):
    import struct
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt " and b[36:40] == b"data"
    fmt, ch, sr = struct.unpack("<HHI", b[20:28])
    assert (fmt, ch) == (3, 1)
    return np.frombuffer(b[44:], dtype="<f4"), sr


def test_render_ragged_batch_matches_oracle(nam_lib, oracle):
    """nam_hip_batch_render_f32: whole signals of different lengths in one resident launch == the 64-frame
    block loop of tools/render.cpp:163-197 run per stream."""
    nam = nam_lib
    lens = [64 * 9 + 5, 64 * 3, 1, 700]
    x = stream_bank(len(lens), max(lens), seed=41)
    for name, ft in (("wavenet_a1_standard", True), ("wavenet_condition_dsp", False), ("lstm", True)):
        model = nam.get_dsp(model_path(name), fast_tanh=ft)
        b = model.batch(len(lens), 64)
        b.Reset(prewarm=True)
        ys = b.render([x[s, :n] for s, n in enumerate(lens)])
        for s, n in enumerate(lens):
            r = _oracle_run(oracle, name, x[s, :n], 64, ft)
            assert ys[s].shape == r.shape == (model.NumOutputChannels(), n)
            assert float(np.max(np.abs(r - ys[s]))) <= _tol(ft) * max(1.0, float(np.max(np.abs(r)))), (name, s)
        b.close()


def test_cpp_render_tool(nam_lib, oracle, tmp_path):
    """cpp/tools/render: the reference's command line (tools/render.cpp:97-110) and the batch form, on the
    reference's example audio; output = mono float32 WAV of channel 0."""
    import os
    import subprocess
    from conftest import ROOT
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cpp")], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "cpp", "tools", "render")
    wav = os.path.join(ROOT, "tests", "golden", "audio", "input.wav")
    x, sr = _read_f32_wav_any(wav)
    out1 = str(tmp_path / "out.wav")
    r = subprocess.run([exe, model_path("wavenet_a1_standard"), wav, out1], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    y, sr_out = _read_f32_wav(out1)
    assert sr_out == 48000 and len(y) == len(x)
    ref = _oracle_run(oracle, "wavenet_a1_standard", x, 64, False)[0]  # the tool leaves fast tanh off, like the reference's render
    assert float(np.max(np.abs(ref - y))) <= 1e-4
    # batch form: three files of different lengths through the slimmable model at a reduced width
    files = []
    for i, n in enumerate((96000, 12345, 64)):
        p = str(tmp_path / f"in{i}.wav")
        _write_f32_wav(p, x[:n], 48000)
        files.append(p)
    outdir = str(tmp_path / "rendered")
    r = subprocess.run([exe, "--slim", "0.34", model_path("slimmable_wavenet"), "--batch", outdir] + files, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for i, n in enumerate((96000, 12345, 64)):
        y, _ = _read_f32_wav(os.path.join(outdir, f"in{i}.wav"))
        ref = _oracle_run(oracle, "slimmable_wavenet", x[:n], 64, False, ratio=0.34)[0]
        assert len(y) == n and float(np.max(np.abs(ref - y))) <= 1e-4
    # error paths of the reference tool
    bad = subprocess.run([exe, model_path("wavenet"), wav, out1, "extra"], capture_output=True, text=True)
    assert bad.returncode == 1 and "Usage" in bad.stderr
    bad = subprocess.run([exe, "--slim", "0.5", model_path("wavenet"), wav, out1], capture_output=True, text=True)
    assert bad.returncode == 1 and "SlimmableModel" in bad.stderr
    p44 = str(tmp_path / "sr44.wav")
    _write_f32_wav(p44, x[:100], 44100)
    bad = subprocess.run([exe, model_path("lstm"), p44, out1], capture_output=True, text=True)
    assert bad.returncode == 1 and "does not match model expected rate" in bad.stderr


def _write_f32_wav(path, x, sr):
    import struct
    data = np.asarray(x, dtype="<f4").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 1, sr, sr * 4, 4, 32)
    open(path, "wb").write(hdr + b"data" + struct.pack("<I", len(data)) + data)


def _read_f32_wav_any(path):
    """the 24-bit PCM example file, decoded independently of the C++ reader"""
    import struct
    b = open(path, "rb").read()
    fmt, ch, sr, _, _, bits = struct.unpack("<HHIIHH", b[20:36])
    assert (fmt, ch, bits) == (1, 1, 24)
    i = b.index(b"data")
    n = struct.unpack("<I", b[i + 4:i + 8])[0]
    raw = np.frombuffer(b[i + 8:i + 8 + n], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
    v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
    v = np.where(v >= 1 << 23, v - (1 << 24), v)
    return (v / 8388608.0).astype(np.float32), sr


@pytest.mark.parametrize("name", ["A2", "slimmable_container"])
def test_container_mixed_submodels_per_stream(nam_lib, oracle, name):
    """SlimmableContainer on the device: streams of one batch run different submodels (LSTM / small WaveNet /
    A1-standard WaveNet for slimmable_container.nam; A2-Lite / A2-Full for A2.nam), chosen per stream with the
    container's max_value rule (container.cpp:103-115)."""
    nam = nam_lib
    ratios = [1.0, 0.0, 0.4, 0.7, 0.2, 0.9]
    n = 64 * 4
    x = stream_bank(len(ratios), n, seed=51)
    model = nam.get_dsp(model_path(name), fast_tanh=True)
    assert model.architecture == "SlimmableContainer"
    b = model.batch(len(ratios), 64)
    b.Reset(prewarm=True)
    for r in sorted(set(ratios)):
        b.SetSlimmableSize(r, [s for s, v in enumerate(ratios) if v == r])
    y = b.process_stream(x, 64)
    for s, r in enumerate(ratios):
        ref = _oracle_run(oracle, name, x[s], 64, True, ratio=r)
        assert float(np.max(np.abs(ref - y[s]))) <= 5e-5 * max(1.0, float(np.max(np.abs(ref)))), (name, s, r)
    b.close()


def test_container_switching_back_and_forth(nam_lib, oracle):
    """A stream that leaves a submodel and comes back: WaveNet submodels restart from Reset + prewarm, the LSTM
    submodel keeps the state it had (the reference's LSTM has no buffers to clear) and is prewarmed again —
    container.cpp:117-139 with each submodel's own Reset."""
    nam = nam_lib
    name = "slimmable_container"
    schedule = [1.0, 0.1, 0.5, 0.1, 1.0, 0.1]
    seg = 64 * 2
    x = stream_bank(2, seg * len(schedule), seed=52)
    model = nam.get_dsp(model_path(name), fast_tanh=True)
    b = model.batch(2, 64)
    b.Reset(prewarm=True)
    ref = oracle.get_dsp(model_path(name), fast_tanh=True)
    ref.Reset(48000.0, 64)
    for i, r in enumerate(schedule):
        b.SetSlimmableSize(r, [1])  # stream 0 stays on the default submodel throughout
        ref.SetSlimmableSize(r)
        xs = x[:, i * seg:(i + 1) * seg]
        y = b.process_stream(xs, 64)
        want = ref.process_stream(xs[1], 64)
        assert float(np.max(np.abs(want - y[1]))) <= 5e-5, (i, r)
    full = _oracle_run(oracle, name, x[0], 64, True)
    b2 = model.batch(1, 64)
    b2.Reset(prewarm=True)
    np.testing.assert_allclose(b2.process_stream(x[:1], 64)[0], full, atol=5e-5)
    b.close()
    b2.close()


@pytest.mark.parametrize("name,in_ch", [("synth_posthead", 1), ("synth_multich", 3)])
@pytest.mark.parametrize("fast_tanh", [True, False])
def test_post_stack_head_and_multichannel_io(nam_lib, oracle, name, in_ch, fast_tanh):
    """a post-stack head (activation -> Conv1D chain after the layer arrays, model.cpp:21-103,
    854-883; two output channels) and a 3-in / 2-out model (test_real_time_safe.cpp:1069). Planar
    [stream][channel][frame] I/O through process(), the f64 API, and the ragged render."""
    nam = nam_lib
    n_streams, n = 4, 64 * 3 + 9
    rng = np.random.default_rng(61)
    x = rng.uniform(-0.5, 0.5, (n_streams, in_ch, n)).astype(np.float32)
    model = nam.get_dsp(model_path(name), fast_tanh=fast_tanh)
    assert (model.NumInputChannels(), model.NumOutputChannels()) == (in_ch, 2)
    refs = []
    for s in range(n_streams):
        r = oracle.get_dsp(model_path(name), fast_tanh=fast_tanh)
        r.Reset(48000.0, 64)
        refs.append(r.process_stream(x[s], 64))
    b = model.batch(n_streams, 64)
    b.Reset(prewarm=True)
    y = b.process_stream(x, 64)
    b.Reset(prewarm=True)
    y64 = b.process_stream(x.astype(np.float64), 64)
    b.Reset(prewarm=True)
    lens = [n, 100, 64, 1]
    yr = b.render([x[s, :, :m] for s, m in enumerate(lens)])
    for s in range(n_streams):
        assert y[s].shape == refs[s].shape == (2, n)
        assert float(np.max(np.abs(refs[s] - y[s]))) <= _tol(fast_tanh)
        assert float(np.max(np.abs(refs[s] - y64[s]))) <= _tol(fast_tanh)
        assert float(np.max(np.abs(refs[s][:, :lens[s]] - yr[s]))) <= _tol(fast_tanh)
    b.close()


@pytest.mark.parametrize("name,in_ch", [("lstm", 1), ("synth_lstm_h18x2", 1), ("synth_lstm_io", 2), ("synth_lstm_h10x2", 1),
                                        ("synth_lstm_h4x2", 1), ("synth_lstm_h2io", 2), ("synth_lstm_h32", 1),
                                        ("synth_lstm_h24x2io", 2)])
@pytest.mark.parametrize("kernel", ["auto", "mfma", "lanes"])
def test_lstm_kernels_match_oracle(nam_lib, oracle, name, in_ch, kernel):
    """The LSTM kernels — gate row per lane (hidden <= 4: AUTO for lstm.nam and the two small fixtures), two gate rows per
    lane and one stream per wavefront (5 .. 32 units: AUTO for the 8-, 10- and 18-unit fixtures; 10 and 18 are padded to
    12 / 20), matrix-core (16 streams per wavefront: forced), lanes-are-streams (GENERIC) — against the oracle: partial
    wavefronts / rows, padding units, several unit tiles with a ragged last one, two layers, 2-in / 3-out."""
    nam = nam_lib
    n_streams, n = 37, 64 * 3 + 11
    rng = np.random.default_rng(71)
    x = rng.uniform(-0.5, 0.5, (n_streams, in_ch, n)).astype(np.float32)
    for fast_tanh in (True, False):
        model = nam.get_dsp(model_path(name), fast_tanh=fast_tanh)
        b = model.batch(n_streams, 64)
        if kernel == "lanes":
            b.set_kernel(nam.KERNEL_GENERIC)
        elif kernel == "mfma":
            b.set_kernel(nam.KERNEL_A1_MFMA)
        small = name in ("lstm", "synth_lstm_h4x2", "synth_lstm_h2io")
        want = {"lanes": "nam_lstm_kernel", "mfma": "nam_lstm_mfma", "auto": "nam_lstm_row_kernel" if small else "nam_lstm_wide_kernel"}[kernel]
        assert b.kernel_name().startswith(want), (b.kernel_name(), want)
        b.Reset(prewarm=True)
        y = b.process_stream(x, 64)
        for s in (0, 1, 2, 3, 4, 15, 16, 31, 32, 35, 36):
            r = oracle.get_dsp(model_path(name), fast_tanh=fast_tanh)
            r.Reset(48000.0, 64)
            ref = r.process_stream(x[s], 64)
            assert float(np.max(np.abs(ref - y[s]))) <= _tol(fast_tanh) * max(1.0, float(np.max(np.abs(ref)))), (name, kernel, s)
        b.close()


def test_device_matches_the_reference_library(nam_lib):
    """The HIP path against the reference's own sources (oracle/_ref/libnam_ref.so, built from /root/reference
    where it exists and shipped prebuilt to the GPU box) — no restatement in between."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nam_ref
    if not os.path.exists(nam_ref.LIB):
        pytest.skip("oracle/_ref/libnam_ref.so was not built (needs /root/reference at build time)")
    nam = nam_lib
    x = stream_bank(3, 64 * 5 + 3, seed=81)
    for name, ft in (("wavenet_a1_standard", True), ("wavenet_a1_standard", False), ("wavenet_a2_max", False), ("A2", True),
                     ("slimmable_container", True), ("lstm", True), ("synth_a1_c8", True)):
        model = nam.get_dsp(model_path(name), fast_tanh=ft)
        b = model.batch(3, 64)
        b.Reset(prewarm=True)
        y = b.process_stream(x, 64)
        for s in range(3):
            ref = nam_ref.get_dsp(model_path(name), ft)
            ref.Reset(48000.0, 64)
            r = ref.process_stream(x[s], 64)
            assert float(np.max(np.abs(r - y[s]))) <= _tol(ft) * max(1.0, float(np.max(np.abs(r)))), (name, ft, s)
        b.close()


def test_edge_cases_of_the_abi(nam_lib, oracle):
    """Empty calls, big buffers, repeated resets, two batches on one model, many streams, zero-length signals."""
    nam = nam_lib
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    x = stream_bank(2, 4096, seed=91)
    # (a) zero frames is a no-op; max_frames 4096 in one call == 64-frame calls (prewarm lengths differ, the
    #     silence steady state does not)
    b = model.batch(2, 4096)
    b.Reset(prewarm=True)
    assert b.process(x[:, :0]).shape == (2, 1, 0)
    y_big = b.process(x)
    # (b) a second batch on the same model is independent of the first
    b2 = model.batch(2, 64)
    b2.Reset(prewarm=True)
    y_small = b2.process_stream(x, 64)
    assert float(np.max(np.abs(y_big - y_small))) <= 1e-5
    # (c) Reset again restores the same start state; Reset without prewarm starts from silence-free zero state
    b2.Reset(prewarm=True)
    np.testing.assert_array_equal(b2.process_stream(x[:, :256], 64), y_small[:, :, :256])
    b2.Reset(prewarm=False)
    y_np = b2.process_stream(x[:, :256], 64)
    ref = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    ref.Reset(48000.0, 64, prewarm=False)
    assert float(np.max(np.abs(ref.process_stream(x[0, :256], 64) - y_np[0]))) <= 5e-5
    b.close()
    b2.close()
    # (d) more streams than CUs x 16, odd count, tiny model on every kernel family
    for name in ("wavenet", "lstm", "synth_a1_c8"):
        m = nam.get_dsp(model_path(name), fast_tanh=True)
        n = 5003
        xs = stream_bank(n, 130, seed=92)
        bb = m.batch(n, 64)
        bb.Reset(prewarm=True)
        ys = bb.process_stream(xs, 64)
        for s in (0, 2501, n - 1):
            r = _oracle_run(oracle, name, xs[s], 64, True)
            assert float(np.max(np.abs(r - ys[s]))) <= 5e-5 * max(1.0, float(np.max(np.abs(r)))), (name, s)
        bb.close()
    # (e) render: a zero-length signal next to real ones
    m = nam.get_dsp(model_path("wavenet"), fast_tanh=False)
    bb = m.batch(3, 64)
    bb.Reset(prewarm=True)
    outs = bb.render([x[0, :100], x[0, :0], x[1, :7]])
    assert [o.shape for o in outs] == [(1, 100), (1, 0), (1, 7)]
    r = _oracle_run(oracle, "wavenet", x[1, :7], 64, False)
    assert float(np.max(np.abs(r - outs[2]))) <= 1e-4
    bb.close()


def test_mfma_kernel_is_deterministic_and_launch_shape_independent(nam_lib):
    """The same audio rendered twice with one launch per block and twice as one resident launch: all four results
    bit-identical (any unsynchronised LDS / ring traffic between the compute and mover wavefronts would show up as
    run-to-run differences); 700 streams = several workgroups per CU."""
    nam = nam_lib
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    n_streams = 700
    x = stream_bank(n_streams, 64 * 30 + 13, seed=123)
    outs = []
    for rep in range(4):
        b = model.batch(n_streams, 64)
        b.set_kernel(nam.KERNEL_A1_MFMA)
        b.Reset(prewarm=True)
        outs.append(b.process_stream(x, 64) if rep % 2 == 0 else np.stack(b.render(list(x))))
        b.close()
    for y in outs[1:]:
        np.testing.assert_array_equal(outs[0], y)


# ---- round 2 pins -------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n_streams", [256, 700])
def test_headline_config_every_stream_matches_oracle(nam_lib, oracle, n_streams):
    """BASELINE.json configs[1] exactly — wavenet_a1_standard, 256 concurrent streams, buffer 64, fast tanh (the
    benchmodel default), one launch per buffer, AUTO kernel — EVERY stream against the oracle over 8 buffers; and 700
    streams (a second / third workgroup per CU), every stream as well."""
    nam = nam_lib
    block, n = 64, 64 * 8
    x = stream_bank(n_streams, n, seed=2024)
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    b = model.batch(n_streams, block)
    b.Reset(prewarm=True)
    y = b.process_stream(x, block)
    b.close()
    ref = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    worst = 0.0
    for s in range(n_streams):
        ref.Reset(48000.0, block)
        r = ref.process_stream(x[s], block)
        err = float(np.max(np.abs(r - y[s])))
        worst = max(worst, err)
        assert err <= 5e-5, (n_streams, s, err)  # ABSOLUTE, as the reference's own bound is written (test_a2_fast.cpp:296-298)
    print(f"headline parity: {n_streams} streams x {n} frames, worst max-abs error {worst:.3e}")


@pytest.mark.parametrize("ratio,channels", [(0.0, 3), (1.0, 8)])
@pytest.mark.parametrize("kernel", ["a1_mfma", "a1", "generic"])
def test_a2_matches_the_reference_fast_path(nam_lib, ratio, channels, kernel):
    """A2.nam against what the reference ACTUALLY runs for it by default: wavenet/a2_fast.cpp (NAM_ENABLE_A2_FAST,
    CMakeLists.txt:58; dispatch model.cpp:1317), built unmodified into oracle/_ref/libnam_ref_a2fast.so. Protocol of
    the reference's own A/B test (tools/test/test_a2_fast.cpp:109-128,272-300): two-tone input, 2,048 frames, block
    sizes 64 and 256, Reset (with prewarm) before each run, max-abs tolerance 5e-5. A2-Lite (3 channels) and A2-Full
    (8); the K-tap MFMA kernel (Full), the VALU kernel and the op-program interpreter."""
    import nam_ref
    if not nam_ref.available():
        pytest.skip("prebuilt oracle/_ref not available")
    nam = nam_lib
    x = two_tone(2048)
    model = nam.get_dsp(model_path("A2"), fast_tanh=False)
    for block in (64, 256):
        ref = nam_ref.get_dsp(model_path("A2"), fast_tanh=False, a2_fast=True)
        ref.Reset(48000.0, block)
        ref.SetSlimmableSize(ratio)
        r = ref.process_stream(x, block)[0]
        b = model.batch(2, block)
        b.set_kernel({"a1_mfma": nam.KERNEL_A1_MFMA, "a1": nam.KERNEL_A1, "generic": nam.KERNEL_GENERIC}[kernel])
        b.Reset(prewarm=True)
        b.SetSlimmableSize(ratio)
        y = b.process_stream(np.stack([x, x]), block)[:, 0, :]
        b.close()
        for s in range(2):
            err = float(np.max(np.abs(r - y[s])))
            assert err <= 5e-5, (channels, kernel, block, s, err)


@pytest.mark.parametrize("name,luts,fast_tanh", [
    ("wavenet", {"Tanh": (-5.0, 5.0, 1024)}, False),
    ("wavenet", {"Tanh": (-5.0, 5.0, 1024)}, True),  # a table wins over fast tanh
    ("wavenet_a1_standard", {"Tanh": (-4.0, 4.0, 4096)}, False),
    ("wavenet_a2_max", {"Sigmoid": (-8.0, 8.0, 1024), "SiLU": (-6.0, 6.0, 333)}, False),  # gated / blended secondaries too
])
def test_lookup_table_activations_match_oracle(nam_lib, oracle, name, luts, fast_tanh):
    """FastLUTActivation (activations.h:371-422) on the device: the table is built on the host exactly as the reference
    builds it, the kernel clamps / indexes / interpolates (device_common.h: d_lut). The oracle it is compared with is
    bit-exact with the reference build for the same tables (tests/test_reference_build.py)."""
    nam = nam_lib
    n_streams, block, n = 3, 64, 64 * 5 + 9
    x = stream_bank(n_streams, n, seed=77)
    model = nam.get_dsp(model_path(name), fast_tanh=fast_tanh, luts=luts)
    b = model.batch(n_streams, block)
    b.Reset(prewarm=True)
    y = b.process_stream(x, block)
    b.close()
    for s in range(n_streams):
        ref = oracle.get_dsp(model_path(name), fast_tanh=fast_tanh, luts=luts)
        ref.Reset(48000.0, block)
        r = ref.process_stream(x[s], block)
        assert float(np.max(np.abs(r - y[s]))) <= 1e-4 * max(1.0, float(np.max(np.abs(r)))), (name, s)


def test_leaky_hardtanh_matches_oracle(nam_lib, oracle):
    """LeakyHardtanh (activations.h:75-89) on the device: object form with four parameters, string form (registry
    defaults) and the alternative spelling; the fixture drives every array past both knees."""
    nam = nam_lib
    n_streams, block, n = 4, 64, 64 * 4 + 31
    x = stream_bank(n_streams, n, seed=78)
    for fast_tanh in (False, True):
        model = nam.get_dsp(model_path("synth_leakyhardtanh"), fast_tanh=fast_tanh)
        b = model.batch(n_streams, block)
        b.Reset(prewarm=True)
        y = b.process_stream(x, block)
        b.close()
        for s in range(n_streams):
            r = _oracle_run(oracle, "synth_leakyhardtanh", x[s], block, fast_tanh)
            assert float(np.max(np.abs(r - y[s]))) <= 5e-5 * max(1.0, float(np.max(np.abs(r)))), s


@pytest.mark.parametrize("name", ["synth_a1_lite", "synth_a1_c14"])
def test_generic_and_padded_a1_kernels_do_not_share_state(nam_lib, oracle, name):
    """Models whose A1 kernels run on zero-padded channels (6 -> 8, 14 -> 16, 10 -> 12) keep two ring layouts: the op
    program's [R][C] and the A1 kernels' [R][C_padded]. Switching family mid-stream must be refused (it used to read
    rings written in the other layout and produce wrong audio silently); after a Reset either family runs and matches
    the oracle, and unpadded models still switch freely."""
    nam = nam_lib
    n_streams, block = 3, 64
    x = stream_bank(n_streams, 64 * 4, seed=91)
    model = nam.get_dsp(model_path(name), fast_tanh=True)
    b = model.batch(n_streams, block)
    b.Reset(prewarm=True)  # AUTO: the MFMA kernel on the padded layout
    assert b.get_kernel() == nam.KERNEL_A1_MFMA
    with pytest.raises(nam.NamHipError):
        b.set_kernel(nam.KERNEL_GENERIC)
    assert b.get_kernel() == nam.KERNEL_A1_MFMA
    b.set_kernel(nam.KERNEL_A1)  # same family: allowed, bit-compatible state
    y0 = b.process(x[:, :64])
    b.Reset(prewarm=False)  # zeroed state: either layout may follow
    b.set_kernel(nam.KERNEL_GENERIC)
    b.Reset(prewarm=True)
    y = b.process_stream(x, block)
    with pytest.raises(nam.NamHipError):
        b.set_kernel(nam.KERNEL_AUTO)
    b.close()
    for s in range(n_streams):
        r = _oracle_run(oracle, name, x[s], block, True)
        assert float(np.max(np.abs(r - y[s]))) <= 5e-5 * max(1.0, float(np.max(np.abs(r))))
        assert float(np.max(np.abs(r[:, :64] - y0[s]))) <= 5e-5 * max(1.0, float(np.max(np.abs(r))))
    # an unpadded model: one layout, free switching (as before)
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    b = model.batch(2, block)
    b.Reset(prewarm=True)
    parts = []
    for i, k in enumerate((nam.KERNEL_A1_MFMA, nam.KERNEL_GENERIC, nam.KERNEL_A1, nam.KERNEL_AUTO)):
        b.set_kernel(k)
        parts.append(b.process(x[:2, 64 * i:64 * (i + 1)]))
    b.close()
    y = np.concatenate(parts, axis=-1)
    for s in range(2):
        r = _oracle_run(oracle, "wavenet_a1_standard", x[s], block, True)
        assert float(np.max(np.abs(r - y[s]))) <= 5e-5


@pytest.mark.parametrize("name", ["wavenet_a1_standard"] + SYNTH_A1)
@pytest.mark.parametrize("fast_tanh", [True, False])
def test_interleaved_mfma_kernel_matches_oracle(nam_lib, oracle, name, fast_tanh):
    """NAM_HIP_KERNEL_A1_IL — the interleaved-frame kernels (frames 4j + w per wave: exchange / DPP / ring-only jobs) exist for the
    official topologies (standard, lite = 12 / 6 padded to 12 / 8, feather), their job tables compiled in: nam_a1_p2_kernel for a
    buffer, the pipelines for longer launches; every other K = 3 topology (13, 12, 15 layers, other dilations: the descriptor-driven
    form was retired in round 5) falls back to the wave-specialised nam_a1_mfma_kernel. One launch per buffer with a ragged
    tail, one multi-block launch, and alternating with the wave-specialised MFMA kernel and the VALU kernel on the same
    state between buffers."""
    nam = nam_lib
    n_streams, block, n = 3, 64, 64 * 6 + 17
    x = stream_bank(n_streams, n, seed=131)
    model = nam.get_dsp(model_path(name), fast_tanh=fast_tanh)
    p2 = bool(model.info.has_a1_kernel & 8)
    assert p2 == (name in ("wavenet_a1_standard", "synth_a1_lite", "synth_a1_feather"))
    want_name = "nam_a1_p2_kernel" if p2 else "nam_a1_mfma_kernel"
    generic = False
    for mode, max_frames in (("blocks", block), ("one_launch", 512)):
        refs = [_oracle_run(oracle, name, x[s], max_frames, fast_tanh) for s in range(n_streams)]
        b = model.batch(n_streams, max_frames)
        b.set_kernel(nam.KERNEL_A1_IL)
        assert b.get_kernel() == (nam.KERNEL_A1_IL if p2 else nam.KERNEL_A1_MFMA) and b.kernel_name() == want_name
        if p2 and not generic:  # launches of more than one block: the pipelined forms
            assert b.kernel_name(512) == "nam_a1_q_kernel"  # (lite and feather: zero-padded to 16 / 8 for it)
        b.Reset(prewarm=True)
        y = b.process_stream(x, max_frames)
        b.close()
        for s in range(n_streams):
            scale = max(1.0, float(np.max(np.abs(refs[s]))))
            err = float(np.max(np.abs(refs[s] - y[s])))
            assert err <= _tol(fast_tanh) * scale, (name, mode, s, err, scale)
    # same rings, same write positions: the three A1 kernels are interchangeable between buffers
    b = model.batch(n_streams, block)
    b.Reset(prewarm=True)
    parts = []
    for i, k0 in enumerate(range(0, 64 * 6, 64)):
        b.set_kernel((nam.KERNEL_A1_IL, nam.KERNEL_A1_MFMA, nam.KERNEL_A1_IL, nam.KERNEL_A1)[i % 4])
        parts.append(b.process(x[:, k0:k0 + 64]))
    y = np.concatenate(parts, axis=-1)
    b.close()
    for s in range(n_streams):
        r = _oracle_run(oracle, name, x[s, :64 * 6], block, fast_tanh)
        assert float(np.max(np.abs(r - y[s]))) <= _tol(fast_tanh) * max(1.0, float(np.max(np.abs(r)))), (name, "switch", s)


def test_interleaved_mfma_kernel_headline_shape_and_determinism(nam_lib, oracle):
    """256 streams x 8 buffers on nam_a1_p2_kernel, every stream against the oracle; the same audio as block launches and
    as one resident launch (twice each) is bit-identical: any unsynchronised hand-off (loader progress word, LDS
    exchange windows, ring rows read back across blocks) would show up as run-to-run differences; 700 streams = more
    workgroups than CUs."""
    nam = nam_lib
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    x = stream_bank(256, 64 * 8, seed=2025)
    b = model.batch(256, 64)
    b.set_kernel(nam.KERNEL_A1_IL)
    b.Reset(prewarm=True)
    y = b.process_stream(x, 64)
    b.close()
    ref = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    for s in range(256):
        ref.Reset(48000.0, 64)
        r = ref.process_stream(x[s], 64)
        assert float(np.max(np.abs(r - y[s]))) <= 5e-5, s
    x = stream_bank(700, 64 * 30 + 13, seed=123)
    outs = []
    for rep in range(4):
        b = model.batch(700, 64)
        b.set_kernel(nam.KERNEL_A1_IL)
        b.Reset(prewarm=True)
        outs.append(b.process_stream(x, 64) if rep % 2 == 0 else np.stack(b.render(list(x))))
        b.close()
    # block launches (nam_a1_p2_kernel) and resident launches (nam_a1_p4_kernel: the same blocks pipelined through three
    # wave sets, sums associated differently) each reproduce themselves bit for bit and agree with each other to rounding
    np.testing.assert_array_equal(outs[0], outs[2])
    np.testing.assert_array_equal(outs[1], outs[3])
    assert float(np.max(np.abs(outs[0] - outs[1]))) <= 5e-6 * max(1.0, float(np.max(np.abs(outs[0]))))


def test_persistent_block_mode_matches_oracle(nam_lib, oracle):
    """Persistent block mode (nam_hip_batch_set_persistent): one resident launch of nam_a1_p2_kernel consumes one
    doorbell per 64-frame buffer. Device-pointer calls walking a resident window, then the blocking host path (fixed
    staging buffers: a new session), a call that is not a 64-frame buffer (falls back to a launch and back), Reset in
    the middle, and leaving the mode: every stream against the oracle throughout."""
    torch = pytest.importorskip("torch")
    nam = nam_lib
    n_streams, block, nb = 7, 64, 12
    x = stream_bank(n_streams, block * nb + 40, seed=321)
    # the kernels that speak the session protocol: nam_a1_p2_kernel (a workgroup per stream), nam_wn_reg_kernel (a
    # wavefront per stream), nam_lstm_row_kernel (a wavefront per four streams: 7 streams = a ragged last workgroup),
    # nam_lstm_wide_kernel (a wavefront per stream)
    for name, kname in (("wavenet_a1_standard", "nam_a1_q_kernel"), ("synth_a1_feather_relu", "nam_a1_p4_kernel"), ("synth_a1_lite", "nam_a1_q_kernel"),
                        ("wavenet_a2_max", "nam_wn_reg_kernel"), ("lstm", "nam_lstm_row_kernel"),
                        ("synth_lstm_h4x2", "nam_lstm_row_kernel"), ("synth_lstm_h18x2", "nam_lstm_wide_kernel")):
        model = nam.get_dsp(model_path(name), fast_tanh=True)
        refs = [_oracle_run(oracle, name, x[s], block, True) for s in range(n_streams)]
        b = model.batch(n_streams, block)
        assert b.set_persistent(True) and b.kernel_name() == kname
        b.Reset(prewarm=True)
        xd = torch.from_numpy(x[:, None, :]).cuda()
        yd = torch.zeros_like(xd)
        T = xd.shape[2]
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        for k in range(6):  # device pointers, walking the resident window: one session, six commands
            b.process_device(xd.data_ptr() + k * block * 4, yd.data_ptr() + k * block * 4, block, T, st.cuda_stream)
        b.flush(st.cuda_stream)
        torch.cuda.synchronize()
        y = yd.cpu().numpy()
        parts = [y[:, :, :6 * block]]
        for k in range(6, 9):  # host path: staging buffers = another window -> the session restarts transparently
            parts.append(b.process(x[:, k * block:(k + 1) * block]))
        parts.append(b.process(x[:, 9 * block:9 * block + 40]))  # 40 frames: an ordinary launch
        y = np.concatenate(parts, axis=-1)
        for s in range(n_streams):
            r = refs[s][:, :9 * block + 40]
            assert float(np.max(np.abs(r - y[s]))) <= 5e-5 * max(1.0, float(np.max(np.abs(r)))), (name, s)
        b.Reset(prewarm=True)  # ends the session, state starts over
        y2 = np.concatenate([b.process(x[:, k * block:(k + 1) * block]) for k in range(3)], axis=-1)
        assert not b.set_persistent(False)
        y3 = b.process(x[:, 3 * block:4 * block])  # launches again
        b.close()
        y2 = np.concatenate([y2, y3], axis=-1)
        for s in range(n_streams):
            assert float(np.max(np.abs(refs[s][:, :4 * block] - y2[s]))) <= 5e-5 * max(1.0, float(np.max(np.abs(refs[s])))), (name, s)


def test_persistent_block_mode_stream_ordered_commands(nam_lib, oracle):
    """Persistent block mode behind a BUSY caller stream: every buffer's input is produced on the stream (a delay plus
    a copy into the window) right before process_device on the same stream, with no host synchronisation in between —
    the command must be ordered behind the producer (the stream-ordered store, not the host's direct one). A
    device-wide synchronize in the middle of the session must return (the launch leaves by itself when the ring is
    empty), and the session carries on afterwards. 256 streams: one workgroup per CU, the headline shape."""
    torch = pytest.importorskip("torch")
    nam = nam_lib
    name, n_streams, block, nb = "wavenet_a1_standard", 256, 64, 10
    x = stream_bank(n_streams, block * nb, seed=77)
    model = nam.get_dsp(model_path(name), fast_tanh=True)
    b = model.batch(n_streams, block)
    assert b.set_persistent(True)
    b.Reset(prewarm=True)
    src = torch.from_numpy(x[:, None, :]).cuda()
    xd = torch.zeros_like(src)
    yd = torch.zeros_like(src)
    T = xd.shape[2]
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for k in range(nb):
            if k % 3 != 2:
                torch.cuda._sleep(200_000)  # ~0.1 ms: the stream is busy when the command is submitted
            xd[:, :, k * block:(k + 1) * block].copy_(src[:, :, k * block:(k + 1) * block], non_blocking=True)
            b.process_device(xd.data_ptr() + k * block * 4, yd.data_ptr() + k * block * 4, block, T, st.cuda_stream)
            if k == 4:
                torch.cuda.synchronize()  # session alive: must not block on it
    b.flush(st.cuda_stream)
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    b.close()
    for s in (0, 1, 100, 255):
        r = _oracle_run(oracle, name, x[s], block, True)
        assert float(np.max(np.abs(r - y[s]))) <= 5e-5 * max(1.0, float(np.max(np.abs(r)))), s
    assert np.isfinite(y).all()


@pytest.mark.parametrize("name,in_ch", [("wavenet_a2_max", 1), ("wavenet_condition_dsp", 1), ("synth_leakyhardtanh", 1),
                                        ("synth_multich", 3), ("wavenet", 1)])
def test_register_resident_wavenet_kernel(nam_lib, oracle, name, in_ch):
    """nam_wn_reg_kernel (every activation in registers, one wavefront per stream; what AUTO runs for wavenet_a2_max):
    64-frame buffers, odd buffer sizes (the history shift by n < 64 frames), one long launch walking many blocks
    (render), a mixed-length ragged render, and the state-layout guard against the op interpreter — all against the
    oracle, both tanh modes where the model has a tanh."""
    nam = nam_lib
    n_streams, n = 6, 64 * 5 + 23
    rng = np.random.default_rng(97)
    x = rng.uniform(-0.6, 0.6, (n_streams, in_ch, n)).astype(np.float32)
    for fast_tanh in (True, False):
        model = nam.get_dsp(model_path(name), fast_tanh=fast_tanh)
        assert model.info.has_a1_kernel & 16
        refs = []
        for s in range(n_streams):
            r = oracle.get_dsp(model_path(name), fast_tanh=fast_tanh)
            r.Reset(48000.0, 64)
            refs.append(r.process_stream(x[s] if in_ch > 1 else x[s, 0], 64))
        b = model.batch(n_streams, 64)
        b.set_kernel(nam.KERNEL_WN_REG)
        assert b.kernel_name() == "nam_wn_reg_kernel"
        xin = x if in_ch > 1 else x[:, 0, :]
        b.Reset(prewarm=True)
        y = b.process_stream(xin, 64)
        b.Reset(prewarm=True)
        y_odd = np.concatenate([b.process(xin[..., a:a + m]) for a, m in
                                [(0, 17), (17, 64), (81, 1), (82, 63), (145, 64), (209, 40), (249, 64), (313, n - 313)]], axis=-1)
        b.Reset(prewarm=True)
        lens = [n, 200, 64, 1, 129, 65]
        yr = b.render([xin[s, ..., :m] for s, m in enumerate(lens)])
        tol = _tol(fast_tanh)
        for s in range(n_streams):
            ref = refs[s].reshape(y[s].shape)
            scale = max(1.0, float(np.max(np.abs(ref))))
            assert float(np.max(np.abs(ref - y[s]))) <= tol * scale, (name, fast_tanh, s)
            # the oracle runs 64-frame calls; other partitions of the same signal agree up to rounding (no PReLU whose
            # slope index depends on the call boundaries reaches this kernel)
            assert float(np.max(np.abs(ref - y_odd[s]))) <= tol * scale, (name, fast_tanh, s, "odd")
            assert float(np.max(np.abs(ref[..., :lens[s]] - np.asarray(yr[s]).reshape(ref[..., :lens[s]].shape)))) <= tol * scale, (name, s, "render")
        # the interpreter keeps rings, this kernel 64-frame histories: switching needs freshly reset state
        with pytest.raises(nam.NamHipError):
            b.set_kernel(nam.KERNEL_GENERIC)
        b.Reset(prewarm=False)
        b.set_kernel(nam.KERNEL_GENERIC)
        b.Reset(prewarm=True)
        yg = b.process_stream(xin, 64)
        for s in range(n_streams):
            assert float(np.max(np.abs(yg[s] - y[s]))) <= 2 * tol * max(1.0, float(np.max(np.abs(yg[s]))))
        b.close()


def test_lds_ring_kernel_mixed_width_batch_one_launch(nam_lib, oracle):
    """Config 5 on nam_wn_reg_kernel: slimmable_wavenet (1..3 channels, dilations up to 512) keeps every layer's
    dilation history in an LDS-resident ring, and the three width groups of a mixed batch share ONE launch (and one
    persistent session). Long enough that the longest ring (1,088 frames) wraps twice. Covered: 64-frame calls (a launch
    fetches only the windows its taps reach), odd partitions, one launch walking the whole signal, the persistent block
    mode with mixed widths, a width change in the middle of a session — every stream against the oracle's slimmed model."""
    torch = pytest.importorskip("torch")
    nam = nam_lib
    name, n_streams, block = "slimmable_wavenet", 11, 64
    n = 64 * 40 + 23
    x = stream_bank(n_streams, n, seed=505)
    ratios = [0.0, 0.34, 0.67, 1.0]
    ratio_of = [ratios[s % 4] for s in range(n_streams)]
    for fast_tanh in (True, False):
        model = nam.get_dsp(model_path(name), fast_tanh=fast_tanh)
        assert model.info.has_a1_kernel & 16
        refs = [_oracle_run(oracle, name, x[s], block, fast_tanh, ratio=ratio_of[s]) for s in range(n_streams)]
        tol = _tol(fast_tanh)

        def check(y, upto, what):
            for s in range(n_streams):
                r = refs[s][..., :upto]
                err = float(np.max(np.abs(r - np.asarray(y[s]).reshape(r.shape))))
                assert err <= tol * max(1.0, float(np.max(np.abs(r)))), (what, fast_tanh, s, err)

        b = model.batch(n_streams, block)
        assert b.kernel_name() == "nam_wn_reg_kernel"  # what AUTO picks for the narrow widths

        def set_widths():
            for i, r in enumerate(ratios):
                b.SetSlimmableSize(r, [s for s in range(n_streams) if s % 4 == i])

        b.Reset(prewarm=True)
        set_widths()
        check(b.process_stream(x, block), n, "64-frame launches")
        b.Reset(prewarm=True)
        lens, rng = [], np.random.default_rng(8)
        while sum(lens) < n:
            lens.append(min(int(rng.integers(1, 65)), n - sum(lens)))
        cuts = np.concatenate([[0], np.cumsum(lens)])
        check(np.concatenate([b.process(x[:, a:z]) for a, z in zip(cuts[:-1], cuts[1:])], axis=-1), n, "odd partition")
        b.Reset(prewarm=True)
        check(b.render(list(x)), n, "one launch")
        # persistent block mode, mixed widths: one session for the three groups
        b.Reset(prewarm=True)
        assert b.set_persistent(True) and b.kernel_name() == "nam_wn_reg_kernel"
        xd = torch.from_numpy(x[:, None, :]).cuda()
        yd = torch.zeros_like(xd)
        T = xd.shape[2]
        nb = n // block
        torch.cuda.synchronize()
        for k in range(nb):
            b.process_device(xd.data_ptr() + k * block * 4, yd.data_ptr() + k * block * 4, block, T)
            if k in (3, 20):
                b.flush()  # the launch leaves, the next buffer starts it again: state through HBM and back
        b.flush()
        torch.cuda.synchronize()
        check(yd.cpu().numpy()[:, 0, :nb * block], nb * block, "persistent")
        b.close()

    # a width change in the middle of a persistent session: the moved stream restarts from Reset + prewarm
    model = nam.get_dsp(model_path(name), fast_tanh=True)
    b = model.batch(4, block)
    assert b.set_persistent(True)
    b.Reset(prewarm=True)
    b.SetSlimmableSize(0.0, [1])
    ref = oracle.get_dsp(model_path(name), fast_tanh=True)
    ref.SetSlimmableSize(0.0)
    ref.Reset(48000.0, block)
    seg = block * 6
    ys, want = [], []
    for i, r in enumerate([0.0, 1.0, 0.5, 0.0]):
        if i:
            b.SetSlimmableSize(r, [1])
            ref.SetSlimmableSize(r)
        xs = x[:4, i * seg:(i + 1) * seg]
        ys.append(b.process_stream(xs, block))
        want.append(ref.process_stream(xs[1], block))
    y = np.concatenate(ys, axis=-1)
    assert float(np.max(np.abs(np.concatenate(want, axis=-1) - y[1]))) <= 5e-5
    full = _oracle_run(oracle, name, x[0, :4 * seg], block, True)
    assert float(np.max(np.abs(full - y[0]))) <= 5e-5
    b.close()


@pytest.mark.parametrize("name", ["wavenet_a1_standard", "wavenet_a2_max", "slimmable_wavenet", "A2"])
def test_prewarm_cache_matches_a_fresh_prewarm(nam_lib, oracle, name):
    """The prewarm cache (conv1d.cpp:151-161, model.cpp:737-775: the reference caches what prewarm leaves in every conv
    and refills from it on later Resets): the first Reset runs the silence and keeps one stream's state, later Resets —
    and per-stream SetSlimmableSize — copy it. Bit-equal to the first run, equal to the oracle, and a Reset without
    prewarm still starts from zero state."""
    nam = nam_lib
    n_streams, block, n = 5, 64, 64 * 4
    x = stream_bank(n_streams, n, seed=61)
    model = nam.get_dsp(model_path(name), fast_tanh=True)
    b = model.batch(n_streams, block)
    b.Reset(prewarm=True)
    y1 = b.process_stream(x, block)  # prewarm ran
    b.Reset(prewarm=True)
    y2 = b.process_stream(x, block)  # state copied from the cache
    np.testing.assert_array_equal(y1, y2)
    b.Reset(prewarm=False)
    y0 = b.process_stream(x, block)
    b.Reset(prewarm=True)
    np.testing.assert_array_equal(y1, b.process_stream(x, block))
    r = _oracle_run(oracle, name, x[2], block, True)
    assert float(np.max(np.abs(r - y1[2]))) <= 5e-5 * max(1.0, float(np.max(np.abs(r))))
    ref0 = oracle.get_dsp(model_path(name), fast_tanh=True)
    ref0.Reset(48000.0, block, prewarm=False)
    r0 = ref0.process_stream(x[2], block)
    assert float(np.max(np.abs(r0 - y0[2]))) <= 5e-5 * max(1.0, float(np.max(np.abs(r0))))
    if model.is_slimmable:
        # stream 1 leaves and comes back: the second visit of a width is served from that width's cache
        for ratio in (0.0, 1.0, 0.0, 1.0):
            b.SetSlimmableSize(ratio, [1])
            y = b.process_stream(x, block)
            rr = oracle.get_dsp(model_path(name), fast_tanh=True)
            rr.SetSlimmableSize(ratio)
            rr.Reset(48000.0, block)
            want = rr.process_stream(x[1], block)
            assert float(np.max(np.abs(want - y[1]))) <= 5e-5 * max(1.0, float(np.max(np.abs(want)))), ratio
    b.close()
