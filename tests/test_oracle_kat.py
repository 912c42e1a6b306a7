"""Pins the CPU oracle against the reference's own primitive-level known-answer tests
(re-expressed from tools/test/*.cpp; expected values and tolerances are the reference's)."""
import numpy as np
import pytest

F = np.float32


def _fp(oracle, a):
    return oracle._fptr(np.ascontiguousarray(a, dtype=np.float32))


def conv1d(oracle, in_ch, out_ch, K, bias, dil, weights, x, groups=1, n_calls=1, max_buf=64):
    """x: [in_ch, frames] -> [out_ch, frames]; processed in n_calls equal blocks on one Conv1D."""
    x = np.asarray(x, dtype=F)
    frames = x.shape[1]
    per = frames // n_calls
    xin = np.ascontiguousarray(x.T.reshape(-1))  # column-major (ch x frames)
    out = np.zeros(out_ch * frames, dtype=F)
    w = np.ascontiguousarray(weights, dtype=F)
    rc = oracle.lib().orc_kat_conv1d(in_ch, out_ch, K, int(bias), dil, groups, oracle._fptr(w), len(w),
                                     oracle._fptr(xin), oracle._fptr(out), per, n_calls, max_buf)
    assert rc == 0
    return out.reshape(frames, out_ch).T


def conv1x1(oracle, in_ch, out_ch, bias, weights, x, groups=1):
    x = np.asarray(x, dtype=F)
    frames = x.shape[1]
    xin = np.ascontiguousarray(x.T.reshape(-1))
    out = np.zeros(out_ch * frames, dtype=F)
    w = np.ascontiguousarray(weights, dtype=F)
    assert oracle.lib().orc_kat_conv1x1(in_ch, out_ch, int(bias), groups, oracle._fptr(w), len(w), oracle._fptr(xin),
                                        oracle._fptr(out), frames) == 0
    return out.reshape(frames, out_ch).T


# ---- tools/test/test_conv1d.cpp:161-342 (hand-computed, tol 0.01) ----
def test_conv1d_basic(oracle):
    y = conv1d(oracle, 1, 1, 2, False, 1, [1.0, 2.0], [[1, 2, 3, 4]])
    np.testing.assert_allclose(y[0], [2, 5, 8, 11], atol=0.01)


def test_conv1d_with_bias(oracle):
    y = conv1d(oracle, 1, 1, 2, True, 1, [1.0, 0.0, 5.0], [[2, 3]])
    np.testing.assert_allclose(y[0], [5, 7], atol=0.01)


def test_conv1d_multichannel(oracle):
    y = conv1d(oracle, 2, 3, 1, False, 1, [1, 0, 0, 1, 1, 1], [[1, 3], [2, 4]])
    np.testing.assert_allclose(y[:, 0], [1, 2, 3], atol=0.01)


def test_conv1d_dilation(oracle):
    y = conv1d(oracle, 1, 1, 2, False, 2, [1.0, 2.0], [[1, 2, 3, 4]])
    np.testing.assert_allclose(y[0], [2, 4, 7, 10], atol=0.01)


def test_conv1d_multiple_calls_keep_history(oracle):
    """Ring buffer across Process() calls (test_conv1d.cpp:344ff): two 2-frame calls == one 4-frame call."""
    one = conv1d(oracle, 1, 1, 2, False, 1, [1.0, 2.0], [[1, 2, 3, 4]])
    two = conv1d(oracle, 1, 1, 2, False, 1, [1.0, 2.0], [[1, 2, 3, 4]], n_calls=2)
    np.testing.assert_array_equal(one, two)


# ---- tools/test/test_conv1d.cpp:18-98,939-949: closed-form generator vs naive triple loop, tol 1e-4 ----
@pytest.mark.parametrize("in_ch,out_ch,K,dil", [(4, 8, 6, 3), (4, 1, 16, 1)])
def test_conv1d_matches_closed_form_reference(oracle, in_ch, out_ch, K, dil):
    frames = 23
    W = np.zeros((K, out_ch, in_ch), dtype=F)
    flat = []
    for o in range(out_ch):
        for i in range(in_ch):
            for k in range(K):
                v = F(0.011) * F(o + 1) + F(0.007) * F(i + 1) - F(0.003) * F(k + 1)
                W[k, o, i] = v
                flat.append(v)
    bias = np.array([F(-0.05) + F(0.019) * F(o + 1) for o in range(out_ch)], dtype=F)
    flat += list(bias)
    x = np.zeros((in_ch, frames), dtype=F)
    for f in range(frames):
        for i in range(in_ch):
            x[i, f] = F(0.21) * F(i + 1) - F(0.037) * F(f + 1) + F(0.004) * F((i + 1) * (f + 1))
    expected = np.zeros((out_ch, frames), dtype=np.float64)
    for f in range(frames):
        for o in range(out_ch):
            s = float(bias[o])
            for k in range(K):
                src = f - dil * (K - 1 - k)
                if src < 0:
                    continue
                s += float(np.dot(W[k, o].astype(np.float64), x[:, src].astype(np.float64)))
            expected[o, f] = s
    y = conv1d(oracle, in_ch, out_ch, K, True, dil, flat, x)
    assert np.max(np.abs(y - expected)) < 1e-4


def test_conv1d_grouped_is_block_diagonal(oracle):
    """Grouped conv == dense conv with zero off-diagonal blocks (conv1d.cpp:40-52)."""
    rng = np.random.default_rng(0)
    in_ch, out_ch, K, g = 4, 6, 3, 2
    wg = rng.standard_normal(K * in_ch * out_ch // g + out_ch).astype(F)
    x = rng.standard_normal((in_ch, 40)).astype(F)
    y = conv1d(oracle, in_ch, out_ch, K, True, 2, wg, x, groups=g)
    # expand to dense weights in stream order [out][in][k]
    dense = np.zeros((out_ch, in_ch, K), dtype=F)
    p = 0
    opg, ipg = out_ch // g, in_ch // g
    for gi in range(g):
        for i in range(opg):
            for j in range(ipg):
                for k in range(K):
                    dense[gi * opg + i, gi * ipg + j, k] = wg[p]
                    p += 1
    yd = conv1d(oracle, in_ch, out_ch, K, True, 2, np.concatenate([dense.reshape(-1), wg[p:]]), x)
    np.testing.assert_allclose(y, yd, atol=1e-6)


# ---- tools/test/test_conv_1x1.cpp:19-77,560-574 ----
def test_conv1x1_matches_closed_form_reference(oracle):
    in_ch, out_ch, frames = 6, 5, 17
    W = np.array([[F(0.013) * F(o + 1) - F(0.009) * F(i + 1) for i in range(in_ch)] for o in range(out_ch)], dtype=F)
    b = np.array([F(0.02) * F(o + 1) for o in range(out_ch)], dtype=F)
    x = np.array([[F(0.1) * F(i + 1) - F(0.01) * F(f) for f in range(frames)] for i in range(in_ch)], dtype=F)
    y = conv1x1(oracle, in_ch, out_ch, True, np.concatenate([W.reshape(-1), b]), x)
    exp = W.astype(np.float64) @ x.astype(np.float64) + b[:, None]
    assert np.max(np.abs(y - exp)) < 1e-4


def test_conv1x1_depthwise_equals_diag(oracle):
    """groups == in == out: one weight per channel, same stream order (dsp.cpp:331-345,365-372)."""
    w = np.array([2.0, -1.0, 0.5, 1.0, 10.0, -20.0, 3.0, 0.0], dtype=F)  # 4 weights + 4 biases
    x = np.arange(12, dtype=F).reshape(4, 3)
    y = conv1x1(oracle, 4, 4, True, w, x, groups=4)
    np.testing.assert_allclose(y, x * w[:4, None] + w[4:, None], atol=1e-6)


# ---- tools/test/test_film.cpp:26-120 ----
def _film(oracle, cond_dim, input_dim, shift, weights, x, cond, groups=1):
    frames = x.shape[1]
    out = np.zeros(input_dim * frames, dtype=F)
    w = np.ascontiguousarray(weights, dtype=F)
    assert oracle.lib().orc_kat_film(cond_dim, input_dim, int(shift), groups, oracle._fptr(w), len(w),
                                     oracle._fptr(np.ascontiguousarray(x.T.reshape(-1), dtype=F)),
                                     oracle._fptr(np.ascontiguousarray(cond.T.reshape(-1), dtype=F)),
                                     oracle._fptr(out), frames) == 0
    return out.reshape(frames, input_dim).T


def test_film_bias_only(oracle):
    w = np.zeros(6 * 2 + 6, dtype=F)
    w[12:] = [2.0, -1.0, 0.5, 10.0, -20.0, 3.0]
    x = np.array([[1, 2, 3, 4], [-1, -2, -3, -4], [0.25, 0.5, 0.75, 1.0]], dtype=F)
    cond = np.random.default_rng(1).standard_normal((2, 4)).astype(F)
    y = _film(oracle, 2, 3, True, w, x, cond)
    exp = x * np.array([2.0, -1.0, 0.5], dtype=F)[:, None] + np.array([10.0, -20.0, 3.0], dtype=F)[:, None]
    assert np.max(np.abs(y - exp)) < 1e-6


def test_film_scale_only(oracle):
    w = np.zeros(3 * 2 + 3, dtype=F)
    w[6:] = [2.0, -1.0, 0.5]
    x = np.array([[1, 2, 3, 4], [-1, -2, -3, -4], [0.25, 0.5, 0.75, 1.0]], dtype=F)
    cond = np.ones((2, 4), dtype=F)
    y = _film(oracle, 2, 3, False, w, x, cond)
    assert np.max(np.abs(y - x * np.array([2.0, -1.0, 0.5], dtype=F)[:, None])) < 1e-6


def test_film_uses_condition(oracle):
    """scale = W_top*c + b, shift = W_bottom*c + b (film.h:101,178-182)."""
    rng = np.random.default_rng(2)
    cd, D, n = 3, 2, 5
    W = rng.standard_normal((2 * D, cd)).astype(F)
    b = rng.standard_normal(2 * D).astype(F)
    x = rng.standard_normal((D, n)).astype(F)
    c = rng.standard_normal((cd, n)).astype(F)
    y = _film(oracle, cd, D, True, np.concatenate([W.reshape(-1), b]), x, c)
    ss = W.astype(np.float64) @ c + b[:, None]
    np.testing.assert_allclose(y, x * ss[:D] + ss[D:], atol=1e-5)


# ---- activations: tools/test/test_activations.cpp ----
def _act(oracle, cfg, x):
    x = np.ascontiguousarray(x, dtype=F).copy()
    a = oracle.act_cfg(cfg)
    oracle.lib().orc_kat_activation(oracle._fptr(a), oracle._fptr(x), x.size)
    return x


def test_activation_values(oracle):
    x = np.array([-2.0, -0.5, 0.0, 0.5, 2.0], dtype=F)
    assert _act(oracle, "Fasttanh", [0.0])[0] == 0.0  # test_activations.cpp:21-31
    np.testing.assert_allclose(_act(oracle, "Tanh", x), np.tanh(x), atol=1e-6)
    np.testing.assert_allclose(_act(oracle, "ReLU", x), np.maximum(x, 0))
    np.testing.assert_allclose(_act(oracle, "LeakyReLU", x), np.where(x > 0, x, F(0.01) * x), atol=1e-7)
    np.testing.assert_allclose(_act(oracle, {"type": "LeakyReLU", "negative_slope": 0.2}, x),
                               np.where(x > 0, x, F(0.2) * x), atol=1e-7)
    np.testing.assert_allclose(_act(oracle, "Sigmoid", x), 1 / (1 + np.exp(-x.astype(np.float64))), atol=1e-6)
    np.testing.assert_allclose(_act(oracle, "SiLU", x), x / (1 + np.exp(-x.astype(np.float64))), atol=1e-6)
    np.testing.assert_allclose(_act(oracle, "Softsign", x), x / (1 + np.abs(x)), atol=1e-7)
    np.testing.assert_allclose(_act(oracle, "Hardtanh", x), np.clip(x, -1, 1))
    np.testing.assert_allclose(_act(oracle, "Hardswish", x), x * np.clip(x + 3, 0, 6) / 6, atol=1e-6)
    lht = _act(oracle, {"type": "LeakyHardtanh", "min_val": 0.0, "max_val": 0.9, "min_slope": 0.0, "max_slope": 0.02}, x)
    np.testing.assert_allclose(lht, [0.0, 0.0, 0.0, 0.5, (2.0 - 0.9) * 0.02 + 0.9], atol=1e-6)
    # fast_tanh tracks tanh to ~1e-3 (it is an approximation by construction, activations.h:91-98)
    assert np.max(np.abs(_act(oracle, "Fasttanh", x) - np.tanh(x))) < 5e-3


def test_prelu_per_channel_flat_indexing(oracle):
    """ActivationPReLU::apply(float*, size): slope = slopes[pos % n] on column-major data (activations.h:283-297)."""
    x = -np.ones(6, dtype=F)  # 2 channels x 3 frames, column-major
    y = _act(oracle, {"type": "PReLU", "negative_slopes": [0.1, 0.5]}, x)
    np.testing.assert_allclose(y, [-0.1, -0.5, -0.1, -0.5, -0.1, -0.5], atol=1e-7)


# ---- gating / blending: test_gating_activations.cpp, test_blending_detailed.cpp:26-112 ----
def _gate(oracle, mode, a1, a2, z, B):
    n = z.shape[1]
    buf = np.ascontiguousarray(z.T.reshape(-1), dtype=F).copy()
    oracle.lib().orc_kat_gating(mode, oracle._fptr(oracle.act_cfg(a1)), oracle._fptr(oracle.act_cfg(a2)), B,
                                oracle._fptr(buf), n)
    return buf.reshape(n, 2 * B).T[:B]


def test_gating_identity_sigmoid(oracle):
    z = np.array([[1.0, -1.0, 0.0], [0.5, 0.8, 1.0]], dtype=F)
    y = _gate(oracle, 1, None, "Sigmoid", z, 1)
    np.testing.assert_allclose(y[0], z[0] / (1 + np.exp(-z[1].astype(np.float64))), atol=1e-6)


def test_gating_leaky_relus(oracle):
    z = np.array([[-1.0, 1.0], [-2.0, 0.5]], dtype=F)
    y = _gate(oracle, 1, {"type": "LeakyReLU", "negative_slope": 0.01}, {"type": "LeakyReLU", "negative_slope": 0.05}, z, 1)
    np.testing.assert_allclose(y[0], [(-0.01) * (-0.1), 0.5], atol=1e-7)


def test_blending_linear_and_sigmoid_return_input(oracle):
    z = np.array([[1, 2], [3, 4], [0.5, 0.8], [0.3, 0.6]], dtype=F)
    for a2 in (None, "Sigmoid"):
        y = _gate(oracle, 2, None, a2, z, 2)
        assert np.max(np.abs(y - z[:2])) < 1e-6


def test_blending_uses_pre_activation(oracle):
    assert abs(_gate(oracle, 2, "ReLU", None, np.array([[2.0], [0.5]], dtype=F), 1)[0, 0] - 2.0) < 1e-6
    assert abs(_gate(oracle, 2, "ReLU", None, np.array([[-1.0], [0.5]], dtype=F), 1)[0, 0] + 0.5) < 1e-6


# ---- tools/test/test_wavenet/test_layer.cpp:40-117: gated layer, EXACT equality ----
def test_layer_gated_exact(oracle):
    L = oracle.lib()
    h = L.orc_wavenet_new(1, 0)
    film = np.zeros(24, dtype=np.int32)
    film[2::3] = 1
    assert L.orc_wavenet_add_array(h, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 0, 1, 1, oracle._iptr(film), 1) == 0
    L.orc_wavenet_add_layer(h, 0, 1, 1, oracle._fptr(oracle.act_cfg("ReLU")), 1, oracle._fptr(oracle.act_cfg("Sigmoid")))
    w = np.array([1.0, 1.0, 0.0, 0.0, 1.0, -1.0, 1.0, 0.0], dtype=F)
    x = np.full(4, 0.25, dtype=F)
    out_next, out_head = np.zeros(4, dtype=F), np.zeros(4, dtype=F)
    rc = L.orc_kat_layer(h, oracle._fptr(w), len(w), oracle._fptr(x), oracle._fptr(x.copy()), oracle._fptr(out_next),
                         oracle._fptr(out_head), 4)
    assert rc == 1
    assert (out_next == 0.5).all() and (out_head == 0.25).all()
    L.orc_wavenet_free(h)
