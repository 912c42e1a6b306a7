"""Oracle regression pin: the committed golden vectors (tests/golden/outputs.npz, produced by
tests/golden/make_golden.py) are reproduced by the oracle on this machine."""
import os

import numpy as np
import pytest

from conftest import ROOT, model_path

G = np.load(os.path.join(ROOT, "tests", "golden", "outputs.npz"))
KEYS = [k for k in G.files if k != "input" and "__ft" in k]


@pytest.mark.parametrize("key", KEYS)
def test_oracle_reproduces_golden(oracle, key):
    name, ft = key.split("__ft")
    m = oracle.get_dsp(model_path(name), fast_tanh=bool(int(ft)))
    m.Reset(48000.0, 64)
    y = m.process_stream(G["input"], 64)
    # exact for pure-arithmetic models; a couple of ulp of slack for libm tanhf / expf across glibc builds
    np.testing.assert_allclose(y, G[key], rtol=0, atol=2e-6 * max(1.0, float(np.max(np.abs(G[key])))))


def test_oracle_slimmed_golden(oracle):
    m = oracle.get_dsp(model_path("slimmable_wavenet"))
    for ratio, tag in ((0.0, "w1"), (0.34, "w2")):
        m.SetSlimmableSize(ratio)
        m.Reset(48000.0, 64)
        y = m.process_stream(G["input"], 64)
        np.testing.assert_allclose(y, G[f"slimmable_wavenet__{tag}"], rtol=0, atol=1e-5)


def test_block_size_independence(oracle):
    """A WaveNet is causal-streaming: results do not depend on how the audio is chopped into calls."""
    x = G["input"]
    m = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    m.Reset(48000.0, 64, prewarm=False)
    a = m.process_stream(x, 64)
    m2 = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    m2.Reset(48000.0, 256, prewarm=False)
    b = m2.process_stream(x, 37)
    np.testing.assert_array_equal(a, b)


def test_prewarm_from_cache_equals_fresh_prewarm(oracle):
    """Second Reset refills the rings from the cached steady column (conv1d.cpp:151-161): same output."""
    x = G["input"][:128]
    m = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    m.Reset(48000.0, 64)
    a = m.process_stream(x, 64)
    m.Reset(48000.0, 64)
    b = m.process_stream(x, 64)
    np.testing.assert_array_equal(a, b)


def test_slimmed_large_equals_small_model(oracle):
    """tools/test/test_slimmable_wavenet.cpp:395-503: the width-w slice of the full weights, run as a
    plain WaveNet, equals the slimmable model at that ratio (tol 1e-6)."""
    import json
    with open(model_path("slimmable_wavenet")) as f:
        j = json.load(f)
    sl = oracle.get_dsp(model_path("slimmable_wavenet"))
    for ratio, width in ((0.0, 1), (0.5, 2), (1.0, 3)):
        assert sl.channels_for(ratio) == [width]
        w = sl.slimmed_weights([width])
        small = json.loads(json.dumps(j))
        lc = small["config"]["layers"][0]
        lc["channels"] = width
        del lc["slimmable"]
        small["weights"] = [float(v) for v in w]
        ref = oracle.load_nam_json(small)
        sl.SetSlimmableSize(ratio)
        x = (0.1 * np.sin(0.1 * np.arange(320))).astype(np.float32)  # test_slimmable_wavenet.cpp:478
        sl.Reset(48000.0, 64)
        ref.Reset(48000.0, 64)
        assert np.max(np.abs(sl.process_stream(x, 64) - ref.process_stream(x, 64))) <= 1e-6
