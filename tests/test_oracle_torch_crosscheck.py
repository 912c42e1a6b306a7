"""Independent second implementation: the plain WaveNet path re-done with torch.nn.functional.conv1d
(dilation=..., CPU float32) straight from the .nam JSON, so a bug in the oracle's convolution /
weight-stream walk would not be shared. Covers the A1 family (wavenet.nam, wavenet_a1_standard.nam)."""
import json

import numpy as np
import pytest

from conftest import model_path
from signals import two_tone

torch = pytest.importorskip("torch")
Fn = torch.nn.functional


def torch_wavenet(path, x, use_tanh=torch.tanh):
    """Whole-signal evaluation with zero history (== Reset without prewarm, then process)."""
    j = json.load(open(path))
    cfg = j["config"]
    w = torch.tensor(j["weights"], dtype=torch.float32)
    p = 0

    def take(n):
        nonlocal p
        v = w[p:p + n]
        p += n
        return v

    xin = torch.tensor(x, dtype=torch.float32)[None, None, :]  # [1, 1, T]
    cond = xin
    layer_in, head_in = xin, None
    for lc in cfg["layers"]:
        C, K, cs, ins, H = lc["channels"], lc["kernel_size"], lc["condition_size"], lc["input_size"], lc["head_size"]
        act = {"Tanh": use_tanh, "ReLU": torch.relu}[lc["activation"]]
        h = Fn.conv1d(layer_in, take(C * ins).view(C, ins, 1))
        head = torch.zeros(1, C, h.shape[2]) if head_in is None else head_in
        for d in lc["dilations"]:
            cw, cb = take(C * C * K).view(C, C, K), take(C)
            mw = take(C * cs).view(C, cs, 1)
            lw, lb = take(C * C).view(C, C, 1), take(C)
            z = Fn.conv1d(Fn.pad(h, ((K - 1) * d, 0)), cw, cb, dilation=d) + Fn.conv1d(cond, mw)
            z = act(z)
            head = head + z
            h = h + Fn.conv1d(z, lw, lb)
        hw = take(H * C).view(H, C, 1)
        hb = take(H) if lc["head_bias"] else None
        head_in = Fn.conv1d(head, hw, hb)
        layer_in = h
    head_scale = take(1)
    assert p == len(w)
    return (head_scale * head_in)[0].numpy()


@pytest.mark.parametrize("name", ["wavenet", "wavenet_a1_standard"])
def test_oracle_matches_torch_conv1d(oracle, name):
    x = two_tone(700)
    ref = torch_wavenet(model_path(name), x)
    m = oracle.get_dsp(model_path(name), fast_tanh=False)
    m.Reset(48000.0, 64, prewarm=False)
    y = m.process_stream(x, 64)
    assert y.shape == ref.shape
    assert np.max(np.abs(y - ref)) < 2e-6, float(np.max(np.abs(y - ref)))


def test_oracle_fast_tanh_matches_torch(oracle):
    def fast_tanh(v):
        ax, x2 = v.abs(), v * v
        return (v * (2.45550750702956 + 2.45550750702956 * ax + (0.893229853513558 + 0.821226666969744 * ax) * x2)
                / (2.44506634652299 + (2.44506634652299 + x2) * (v + 0.814642734961073 * v * ax).abs()))

    x = two_tone(300)
    ref = torch_wavenet(model_path("wavenet_a1_standard"), x, use_tanh=fast_tanh)
    m = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    m.Reset(48000.0, 64, prewarm=False)
    assert np.max(np.abs(m.process_stream(x, 64) - ref)) < 2e-6


def test_lstm_matches_torch_lstm_cell(oracle):
    """lstm.nam against torch.nn.LSTMCell math (gate order i,f,g,o — lstm.cpp:44-66)."""
    j = json.load(open(model_path("lstm")))
    H, I = j["config"]["hidden_size"], j["config"]["input_size"]
    w = np.array(j["weights"], dtype=np.float32)
    W = torch.tensor(w[:4 * H * (I + H)]).view(4 * H, I + H)
    b = torch.tensor(w[4 * H * (I + H):4 * H * (I + H) + 4 * H])
    o = 4 * H * (I + H) + 4 * H
    h, c = torch.tensor(w[o:o + H]), torch.tensor(w[o + H:o + 2 * H])
    hw, hb = torch.tensor(w[o + 2 * H:o + 3 * H]), torch.tensor(w[o + 3 * H])
    x = np.random.default_rng(0).uniform(-0.5, 0.5, 200).astype(np.float32)
    ys = []
    for v in x:
        g = W @ torch.cat([torch.tensor([v]), h]) + b
        i_, f_, g_, o_ = g[:H], g[H:2 * H], g[2 * H:3 * H], g[3 * H:]
        c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
        h = torch.sigmoid(o_) * torch.tanh(c)
        ys.append(float(hw @ h + hb))
    m = oracle.get_dsp(model_path("lstm"), fast_tanh=False)
    m.Reset(48000.0, 64, prewarm=False)
    y = m.process_stream(x, 64)[0]
    assert np.max(np.abs(y - np.array(ys, dtype=np.float32))) < 2e-6
