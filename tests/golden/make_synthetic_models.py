"""Writes small synthetic A1-family WaveNet fixtures (seeded random weights) that exercise the MFMA kernel's
variants the shipped example models do not reach: odd layer counts (idle job + padded launch), prefetch depth
5 / 6, 12- and 4-channel arrays (full lane layout, partial quads), two 8-channel arrays (half layout end to
end), mixed activations (run-time activation dispatch), head bias on / off; and single-array models with
per-layer kernel sizes 1..16 and a head rechannel with taps (synth_kt_*: the K-tap MFMA kernel).

    python tests/golden/make_synthetic_models.py      # then tests/golden/make_golden.py

Weight stream order per layer array (NAM/wavenet/model.cpp:152-181,563-569): rechannel [C][in] (no bias);
per layer conv [C][C][K], conv bias [C], input mixin [C][cond], layer1x1 [C][C], its bias [C];
head rechannel [H][C] (+ bias [H] when head_bias); finally head_scale.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

SPECS = {
    # 7 + 6 = 13 layers -> 14 jobs with an idle one; depth 6; 14 % 6 != 0 -> padded tail
    "synth_a1_13": dict(arrays=[(16, [1, 2, 4, 8, 16, 32, 64], "Tanh", False), (8, [1, 2, 4, 8, 16, 32], "Tanh", True)], seed=11),
    # 12 and 4 channels (full layout, partial quads), dilations beyond one block, ReLU; 12 jobs, depth 6
    "synth_a1_c12": dict(arrays=[(12, [128, 256, 512, 1, 2, 4], "ReLU", True), (4, [8, 16, 32, 64, 3, 5], "ReLU", True)], seed=12),
    # two 8-channel arrays: half layout through X0, PRE_HEAD (previous half) and POST_RECH; 10 jobs, depth 5
    "synth_a1_c8": dict(arrays=[(8, [1, 2, 4, 8, 16], "Tanh", False), (8, [32, 64, 128, 256, 512], "Tanh", True)], seed=13),
    # mixed activations -> run-time dispatch; three arrays (16 -> 8 -> 4); 15 layers -> 16 jobs
    # the official "lite" shape: 12 -> 6 channels, ten layers each; 6 is not a multiple of 4 — the plan compiler
    # zero-pads it to 8 for the A1 kernels (plan_a1.cpp: pad_channels_for_mfma)
    "synth_a1_lite": dict(arrays=[(12, [1, 2, 4, 8, 16, 32, 64, 128, 256, 512], "Tanh", False),
                                  (6, [1, 2, 4, 8, 16, 32, 64, 128, 256, 512], "Tanh", True)], seed=15),
    # the official "feather" shape: 8 -> 4 channels, ten layers each (half layout first, then a single channel quad):
    # the third instantiation of the compile-time-topology kernel (nam_a1_p2_kernel<8, 4>)
    "synth_a1_feather": dict(arrays=[(8, [1, 2, 4, 8, 16, 32, 64, 128, 256, 512], "Tanh", False),
                                     (4, [1, 2, 4, 8, 16, 32, 64, 128, 256, 512], "Tanh", True)], seed=17),
    # the feather shape with another activation: nam_a1_q_kernel is compiled for Tanh / Fasttanh, so this one keeps its own
    # width (no padding to 16 / 8: plan_a1.cpp: official_standard_topology) and its pipeline is nam_a1_p4_kernel<8, 4>
    "synth_a1_feather_relu": dict(arrays=[(8, [1, 2, 4, 8, 16, 32, 64, 128, 256, 512], "ReLU", False),
                                          (4, [1, 2, 4, 8, 16, 32, 64, 128, 256, 512], "ReLU", True)], seed=27),
    # the official "nano" shape: 4 -> 2 channels, ten layers each. Too narrow for the matrix-core kernels (2 channels):
    # nam_wn_reg_kernel's plain-layer runs, 68 KB of LDS-resident rings per stream
    "synth_a1_nano": dict(arrays=[(4, [1, 2, 4, 8, 16, 32, 64, 128, 256, 512], "Tanh", False),
                                  (2, [1, 2, 4, 8, 16, 32, 64, 128, 256, 512], "Tanh", True)], seed=18),
    # 14 -> 10 channels (padded to 16 -> 12), Sigmoid: the padded channels carry f(0) = 0.5 against zero weights
    "synth_a1_c14": dict(arrays=[(14, [1, 2, 4, 8, 16, 32], "Sigmoid", True), (10, [64, 128, 256, 512, 1, 2], "Sigmoid", True)], seed=16),
    "synth_a1_mixed": dict(arrays=[(16, [1, 2, 4, 8, 16], "Tanh", False), (8, [32, 64, 128, 1, 2], "ReLU", True),
                                   (4, [4, 8, 16, 32, 64], "Sigmoid", True)], seed=14),
}


def build(name, arrays, seed):
    rng = np.random.default_rng(seed)
    layers, weights = [], []
    n = len(arrays)
    for i, (C, dil, act, hb) in enumerate(arrays):
        in_size = 1 if i == 0 else arrays[i - 1][0]
        head = 1 if i == n - 1 else arrays[i + 1][0]
        K = 3
        layers.append(dict(input_size=in_size, condition_size=1, head_size=head, channels=C, kernel_size=K, dilations=dil,
                           activation=act, gated=False, head_bias=hb))

        def w(shape, fan_in):
            v = rng.standard_normal(shape).astype(np.float32) * np.float32(0.9 / np.sqrt(fan_in))
            weights.extend(v.reshape(-1).tolist())

        w((C, in_size), in_size)
        for _ in dil:
            w((C, C, K), C * K)
            w((C,), 4.0)
            w((C, 1), 1.0)
            w((C, C), C)
            w((C,), 4.0)
        w((head, C), C * len(dil))
        if hb:
            w((head,), 4.0)
    weights.append(0.05)
    model = dict(version="0.5.4", architecture="WaveNet", config=dict(layers=layers, head=None, head_scale=0.05),
                 metadata=dict(name=name, note="synthetic test fixture (seeded random weights)"), weights=weights, sample_rate=48000)
    with open(os.path.join(HERE, "models", name + ".nam"), "w") as f:
        json.dump(model, f)
    return len(weights)


def build_general(name, in_channels, arrays, post_head, seed, gain=0.9):
    """Plain (ungated, no FiLM) WaveNet with an optional post-stack head and any channel counts.
    arrays: (channels, dilations, activation, head_size, head_bias); post_head: dict or None
    (model.cpp:21-103: repeat activation -> Conv1D(kernel_sizes[i]) with bias, `channels` wide inside)."""
    rng = np.random.default_rng(seed)
    layers, weights = [], []

    def w(shape, fan_in):
        v = rng.standard_normal(shape).astype(np.float32) * np.float32(gain / np.sqrt(fan_in))
        weights.extend(v.reshape(-1).tolist())

    K = 3
    for i, (C, dil, act, head, hb) in enumerate(arrays):
        in_size = in_channels if i == 0 else arrays[i - 1][0]
        layers.append(dict(input_size=in_size, condition_size=in_channels, head_size=head, channels=C, kernel_size=K,
                           dilations=dil, activation=act, gated=False, head_bias=hb))
        w((C, in_size), in_size)
        for _ in dil:
            w((C, C, K), C * K)
            w((C,), 4.0)
            w((C, in_channels), in_channels)
            w((C, C), C)
            w((C,), 4.0)
        w((head, C), C * len(dil))
        if hb:
            w((head,), 4.0)
    if post_head is not None:
        cin = arrays[-1][3]
        ks = post_head["kernel_sizes"]
        for i, k in enumerate(ks):
            cout = post_head["out_channels"] if i + 1 == len(ks) else post_head["channels"]
            w((cout, cin, k), cin * k)
            w((cout,), 4.0)
            cin = cout
    weights.append(0.1)
    config = dict(layers=layers, head=post_head, head_scale=0.1)
    if in_channels != 1:
        config["in_channels"] = in_channels
    model = dict(version="0.5.4", architecture="WaveNet", config=config,
                 metadata=dict(name=name, note="synthetic test fixture (seeded random weights)"), weights=weights, sample_rate=48000)
    with open(os.path.join(HERE, "models", name + ".nam"), "w") as f:
        json.dump(model, f)
    return len(weights)


GENERAL = {
    # post-stack head: ReLU -> Conv1D(K=3) -> ReLU -> Conv1D(K=2), 3 -> 5 -> 2 channels: two outputs
    "synth_posthead": dict(in_channels=1, arrays=[(4, [1, 2, 4], "Tanh", 3, True)],
                           post_head=dict(channels=5, out_channels=2, kernel_sizes=[3, 2], activation="ReLU"), seed=21),
    # LeakyHardtanh (activations.h:75-89) in its object form with all four parameters and in its string form (registry
    # defaults -1, 1, 0.01, 0.01), plus the "LeakyHardTanh" spelling the parser also accepts; inputs driven past both knees
    "synth_leakyhardtanh": dict(in_channels=1, arrays=[
        (4, [1, 2, 4], dict(type="LeakyHardtanh", min_val=-0.5, max_val=0.7, min_slope=0.02, max_slope=0.05), 3, True),
        (3, [1, 3], "LeakyHardtanh", 2, True), (2, [2, 5], dict(type="LeakyHardTanh", min_val=-0.2, max_val=0.1), 1, False)],
        post_head=None, seed=23, gain=2.5),
    # 3 inputs / 2 outputs (the shape tools/test/test_real_time_safe.cpp:1069 uses), no post-stack head
    "synth_multich": dict(in_channels=3, arrays=[(4, [1, 3], "ReLU", 2, False), (2, [2, 5], "Tanh", 2, True)], post_head=None, seed=22),
}


# ---- feature-rich WaveNets: gating / blending / FiLM / grouped convs / head1x1 / bottleneck / nested condition_dsp ----
# Schema: NAM/wavenet/model.cpp:913-1276 (parse_config_json); weight order per layer NAM/wavenet/model.cpp:152-181:
# conv (+bias), input_mixin, layer1x1 (+bias)?, head1x1 (+bias)?, then the eight FiLMs in the fixed order conv_pre,
# conv_post, input_mixin_pre, input_mixin_post, activation_pre, activation_post, layer1x1_post, head1x1_post — each a
# Conv1x1(condition -> (shift ? 2 : 1) * D, bias, groups) (NAM/film.h:28-32). The cases follow the reference's own
# feature tests: tools/test/test_wavenet_configurable_gating.cpp:86-264 (per-layer gating modes, secondary
# activations), test_film.cpp:26-480 (scale only / scale + shift, groups), test_wavenet/test_head1x1.cpp,
# test_wavenet/test_condition_processing.cpp (condition_dsp), test_wavenet_gating_compatibility.cpp (legacy "gated").
FILM_KEYS = ["conv_pre_film", "conv_post_film", "input_mixin_pre_film", "input_mixin_post_film", "activation_pre_film",
             "activation_post_film", "layer1x1_post_film", "head1x1_post_film"]


def _divisors(*ns):
    g = int(np.gcd.reduce([int(n) for n in ns]))
    return [d for d in range(1, g + 1) if g % d == 0]


def _random_activation(rng, rows, allow_prelu=True):
    """An activation in one of the JSON forms ActivationConfig::from_json accepts (activations.cpp:59-130).
    PReLU slopes: one per row of the matrix the activation sees (`rows`), or a single slope."""
    names = ["Tanh", "ReLU", "Sigmoid", "Hardtanh", "SiLU", "Softsign", "Hardswish", "LeakyReLU", "LeakyHardtanh", "Fasttanh"]
    if allow_prelu:
        names.append("PReLU")
    t = names[int(rng.integers(len(names)))]
    if t == "LeakyReLU":
        return dict(type="LeakyReLU", negative_slope=round(float(rng.uniform(0.005, 0.3)), 4)) if rng.integers(2) else "LeakyReLU"
    if t == "LeakyHardtanh":
        if rng.integers(3) == 0:
            return "LeakyHardtanh"
        return dict(type="LeakyHardtanh", min_val=round(float(rng.uniform(-0.9, -0.1)), 3), max_val=round(float(rng.uniform(0.1, 0.9)), 3),
                    min_slope=round(float(rng.uniform(0.0, 0.1)), 3), max_slope=round(float(rng.uniform(0.0, 0.1)), 3))
    if t == "PReLU":
        if rows == 1 and rng.integers(2):  # (one slope for several rows trips the reference's channel-count check, activations.h:300-307)
            return dict(type="PReLU", negative_slope=round(float(rng.uniform(0.01, 0.3)), 3))
        return dict(type="PReLU", negative_slopes=[round(float(v), 3) for v in rng.uniform(0.01, 0.3, rows)])
    return dict(type=t) if rng.integers(2) else t


def random_featured_array(rng, input_size, condition_size, head_size=None, shape=None, max_dilation=40, later=False):
    """One layer array with random features. `shape` pins (channels, bottleneck, kernel_size, head1x1 out or 0, gated?) —
    the dimensions nam_wn_reg_kernel instantiates (plan.h: WR_LAYER_SHAPES) — and leaves FiLM sets, activations, groups,
    dilations and the gating flavour random; None draws everything. Returns the JSON layer-array config."""
    n_layers = int(rng.integers(1, 5))
    if shape is not None:
        C, B, K, HO, gated = shape
        kernel_sizes = [K] * n_layers
    else:
        C = int(rng.integers(1, 9))
        B = C if rng.integers(3) else int(rng.integers(1, 9))
        kernel_sizes = [int(rng.choice([1, 2, 2, 3, 3, 3, 4, 5]))] * n_layers if rng.integers(3) else [int(rng.integers(1, 6)) for _ in range(n_layers)]
        HO = int(rng.integers(1, 9)) if rng.integers(2) else 0
        gated = None
        if later:
            # a later array's head accumulator starts as the previous array's head output, whose size must equal this
            # array's channels (model.cpp:476-484, 643-649): head1x1 -> C outputs, or no head1x1 and bottleneck = C
            if HO or B != C:
                HO = C
    dil = [int(rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 13, 16, max_dilation])) for _ in range(n_layers)]
    if gated is None:
        form = int(rng.integers(5))
        modes = (["none"] * n_layers if form == 0 else ["gated"] * n_layers if form == 1 else ["blended"] * n_layers if form == 2
                 else [str(rng.choice(["none", "gated", "blended"])) for _ in range(n_layers)])
    else:
        modes = [str(rng.choice(["gated", "blended"])) for _ in range(n_layers)] if gated else ["none"] * n_layers
    a = dict(input_size=input_size, condition_size=condition_size, channels=C, dilations=dil)
    if B != C or rng.integers(2):
        a["bottleneck"] = B
    if len(set(kernel_sizes)) == 1 and rng.integers(2):
        a["kernel_size"] = kernel_sizes[0]
    else:
        a["kernel_sizes"] = kernel_sizes
    # the activation sees 2B rows through GatingActivation's per-frame apply on the top B, B rows otherwise: B slopes
    a["activation"] = ([_random_activation(rng, B) for _ in range(n_layers)] if rng.integers(2) else _random_activation(rng, B))
    if all(m == "none" for m in modes) and rng.integers(3) == 0:
        a["gated"] = False  # legacy spelling (model.cpp:1164)
    elif all(m == "gated" for m in modes) and rng.integers(3) == 0:
        a["gated"] = True  # legacy: secondary activation Sigmoid
    else:
        a["gating_mode"] = modes[0] if len(set(modes)) == 1 and rng.integers(2) else modes
        if any(m != "none" for m in modes):
            # (a single gating_mode string takes a single secondary activation: model.cpp:1140-1160)
            k = int(rng.integers(3)) if isinstance(a["gating_mode"], list) else int(rng.integers(1, 3))
            if k == 0:
                a["secondary_activation"] = [_random_activation(rng, B, allow_prelu=False) for _ in range(n_layers)]
            elif k == 1:
                a["secondary_activation"] = _random_activation(rng, B, allow_prelu=False)
            # else: absent -> Sigmoid (model.cpp:1111-1113)
    # groups: must divide both sides of every layer's matrix (zc = B or 2B depending on the layer's gating mode)
    a["groups_input"] = int(rng.choice(_divisors(C, B)))
    a["groups_input_mixin"] = int(rng.choice(_divisors(condition_size, B)))
    l1_active = True if B != C else bool(rng.integers(4))
    a["layer1x1"] = dict(active=l1_active, groups=int(rng.choice(_divisors(B, C))))
    if HO:
        a["head1x1"] = dict(active=True, out_channels=HO, groups=int(rng.choice(_divisors(B, HO))))
    elif rng.integers(2):
        a["head1x1"] = dict(active=False, out_channels=C, groups=1)
    film_dims = dict(zip(FILM_KEYS, [C, None, condition_size, None, None, B, C, HO]))
    p_film = float(rng.choice([0.0, 0.3, 0.6, 1.0]))
    for key in FILM_KEYS:
        if key == "layer1x1_post_film" and not l1_active:
            continue
        if key == "head1x1_post_film" and not HO:
            continue
        if rng.uniform() >= p_film:
            if rng.integers(4) == 0:
                a[key] = False if rng.integers(2) else dict(active=False, shift=True, groups=1)
            continue
        shift = bool(rng.integers(2))
        D = film_dims[key]
        # zc-wide FiLMs see B or 2B rows depending on the layer: a group count that divides B divides both
        g = int(rng.choice(_divisors(condition_size, (2 if shift else 1) * (D if D is not None else B))))
        f = dict(active=True, shift=shift, groups=g)
        if rng.integers(4) == 0 and shift and g == 1:
            f = dict()  # every default (model.cpp:1232-1236)
        a[key] = f
    hs = head_size if head_size is not None else int(rng.integers(1, 5))
    if rng.integers(2):
        a["head_size"] = hs
        a["head_bias"] = bool(rng.integers(2))
    else:
        a["head"] = dict(out_channels=hs, kernel_size=int(rng.choice([1, 1, 2, 3])), bias=bool(rng.integers(2)))
        if rng.integers(3) == 0:
            a["head"]["head_dilation"] = int(rng.choice([1, 2, 3]))
    return a


def _arr_dims(a):
    C = a["channels"]
    B = a.get("bottleneck", C)
    n = len(a["dilations"])
    ks = a["kernel_sizes"] if "kernel_sizes" in a else [a["kernel_size"]] * n
    if "gating_mode" in a:
        modes = a["gating_mode"] if isinstance(a["gating_mode"], list) else [a["gating_mode"]] * n
    else:
        modes = ["gated" if a.get("gated", False) else "none"] * n
    h1 = a.get("head1x1", dict(active=False))
    HO = h1["out_channels"] if h1.get("active") else 0
    l1 = a.get("layer1x1", dict(active=True, groups=1))
    if "head" in a:
        hs, hk, hb = a["head"]["out_channels"], a["head"]["kernel_size"], a["head"]["bias"]
    else:
        hs, hk, hb = a["head_size"], 1, a["head_bias"]
    return C, B, n, ks, modes, h1, HO, l1, hs, hk, hb


def head_input_size(a):
    """Rows of the array's head accumulator: head1x1.out_channels if active, else the bottleneck (model.cpp:397-401)."""
    C, B, n, ks, modes, h1, HO, l1, hs, hk, hb = _arr_dims(a)
    return HO if HO else B


def featured_weight_count(arrays, post_head=None):
    """Length of the flat weight stream the reference consumes for these layer arrays (+ post-stack head + head_scale)."""
    total = 0
    for a in arrays:
        C, B, n, ks, modes, h1, HO, l1, hs, hk, hb = _arr_dims(a)
        cond = a["condition_size"]
        total += C * a["input_size"]
        for i in range(n):
            zc = B if modes[i] == "none" else 2 * B
            total += C * zc // a.get("groups_input", 1) * ks[i] + zc
            total += cond * zc // a.get("groups_input_mixin", 1)
            if l1["active"]:
                total += B * C // l1["groups"] + C
            if HO:
                total += B * HO // h1["groups"] + HO
            dims = [C, zc, cond, zc, zc, B, C, HO]
            for key, D in zip(FILM_KEYS, dims):
                f = a.get(key, False)
                if f is False or not f.get("active", True):
                    continue
                m = 2 if f.get("shift", True) else 1
                total += cond * m * D // f.get("groups", 1) + m * D
        total += (HO if HO else B) * hs * hk + (hs if hb else 0)
    if post_head is not None:
        cin = _arr_dims(arrays[-1])[8]
        for i, k in enumerate(post_head["kernel_sizes"]):
            cout = post_head["out_channels"] if i + 1 == len(post_head["kernel_sizes"]) else post_head["channels"]
            total += cout * cin * k + cout
            cin = cout
    return total + 1


# (channels, bottleneck, kernel, head1x1 outputs, gated) by condition size: the layer dimensions kernel_wn_reg.hip
# instantiates with run-time FiLM / activation flags (plan.h: WR_LAYER_SHAPES ids 5..14, 18)
WR_DIMS = {8: [(4, 4, 4, 4, False)], 1: [(3, 6, 2, 6, True), (4, 2, 3, 4, True), (4, 4, 3, 0, False), (3, 3, 3, 0, False),
                                          (2, 2, 3, 0, False), (8, 8, 3, 0, False), (1, 1, 3, 0, False)],
           3: [(3, 3, 3, 0, False), (4, 4, 3, 0, False), (2, 2, 3, 0, False)]}


def random_featured(rng, wr_shapes=False, allow_condition_dsp=True, in_channels=1, out_size=None, depth=0):
    """A whole feature-rich WaveNet .nam (dict). wr_shapes: layer dimensions from WR_DIMS (what nam_wn_reg_kernel
    instantiates); otherwise free dimensions (the op interpreter). A nested condition_dsp (model.cpp:919-922: its own
    weights inside its JSON) with probability 1/2 at depth 0."""
    cond_dsp = None
    cond = in_channels
    if allow_condition_dsp and depth == 0 and rng.integers(2):
        want = 8 if (wr_shapes and rng.integers(2)) else (int(rng.choice([1, 3])) if wr_shapes else int(rng.integers(1, 7)))
        cond_dsp = random_featured(rng, wr_shapes=wr_shapes, allow_condition_dsp=False, in_channels=in_channels, out_size=want, depth=1)
        cond = want
    elif wr_shapes and in_channels == 1 and depth == 0 and rng.integers(4) == 0:
        in_channels = cond = 3  # multi-channel input: condition = the input channels
    n_arr = int(rng.integers(1, 3)) if depth else int(rng.integers(1, 4))
    shapes = None
    if wr_shapes:
        pool = WR_DIMS.get(cond)
        if pool is None:
            raise ValueError(f"no instantiated shapes for condition size {cond}")
        later = [sh for sh in pool if (sh[3] == sh[0] if sh[3] else sh[1] == sh[0])]  # usable behind another array
        shapes = [(pool if i == 0 else later)[int(rng.integers(len(pool if i == 0 else later)))] for i in range(n_arr)]
    arrays = []
    for i in range(n_arr):
        arrays.append(random_featured_array(rng, in_channels if i == 0 else arrays[-1]["channels"], cond,
                                            shape=shapes[i] if shapes else None, later=i > 0))
    # chain the head sizes: array i's head output seeds array i+1's head accumulator (model.cpp:476-484) and must be
    # as wide as array i+1's channels (:643-649)
    for i in range(n_arr):
        hs = arrays[i + 1]["channels"] if i + 1 < n_arr else (out_size if out_size is not None else int(rng.choice([1, 1, 1, 2])))
        if "head" in arrays[i]:
            arrays[i]["head"]["out_channels"] = hs
        else:
            arrays[i]["head_size"] = hs
    n_w = featured_weight_count(arrays)
    scale = float(rng.choice([0.02, 0.1, 0.5]))
    w = (rng.standard_normal(n_w).astype(np.float32) * np.float32(0.35)).tolist()
    w[-1] = scale  # head_scale comes from the last weight (model.cpp:670)
    config = dict(layers=arrays, head=None, head_scale=scale)
    if in_channels != 1:
        config["in_channels"] = in_channels
    if cond_dsp is not None:
        config["condition_dsp"] = cond_dsp
    return dict(version="0.6.0", architecture="WaveNet", config=config,
                metadata=dict(name="featured fuzz model", note="synthetic test fixture (seeded random weights)"), weights=w,
                sample_rate=48000)


POST_HEAD_ACTS = ["ReLU", "Tanh", "SiLU", "Hardtanh", "Sigmoid", "Softsign", dict(type="LeakyReLU", negative_slope=0.05)]


def add_post_head(m, rng):
    """A post-stack head (model.cpp:21-103: activation -> Conv1D(kernel_sizes[i], bias) per entry, `channels` wide inside)
    behind the last array of model dict `m`: its weights go between the arrays' and head_scale (:661-683)."""
    cfg = m["config"]
    last = cfg["layers"][-1]
    cin = last["head"]["out_channels"] if "head" in last else last["head_size"]
    ks = [int(rng.choice([1, 2, 3, 5])) for _ in range(int(rng.integers(1, 4)))]
    head = dict(channels=int(rng.integers(1, 7)), out_channels=int(rng.choice([1, 1, 2, 3])), kernel_sizes=ks,
                activation=POST_HEAD_ACTS[int(rng.integers(len(POST_HEAD_ACTS)))])
    w = []
    for i, k in enumerate(ks):
        cout = head["out_channels"] if i + 1 == len(ks) else head["channels"]
        w += (rng.standard_normal(cout * cin * k).astype(np.float32) * np.float32(0.6 / np.sqrt(cin * k))).tolist()
        w += (rng.standard_normal(cout).astype(np.float32) * np.float32(0.1)).tolist()
        cin = cout
    m["weights"] = m["weights"][:-1] + w + m["weights"][-1:]
    cfg["head"] = head
    return m


def write_featured(path, seed, wr_shapes=False, post_head=False):
    """Seeded feature-rich model -> `path`; returns the model dict. post_head: with a random post-stack head (drawn from
    a generator of its own: the arrays of a seed are the same with and without it)."""
    rng = np.random.default_rng(seed)
    while True:
        try:
            m = random_featured(rng, wr_shapes=wr_shapes)
            break
        except ValueError:
            continue
    if post_head:
        add_post_head(m, np.random.default_rng(seed + 50_000))
    with open(path, "w") as f:
        json.dump(m, f)
    return m


def build_lstm(name, num_layers, input_size, hidden, out_channels, seed):
    """LSTM weight stream (lstm.cpp:9-29, 70-101): per layer W [4H][I+H] row-major, b [4H], h0 [H], c0 [H];
    then head W [out][H], b [out]."""
    rng = np.random.default_rng(seed)
    weights = []

    def w(shape, scale):
        weights.extend((rng.standard_normal(shape).astype(np.float32) * np.float32(scale)).reshape(-1).tolist())

    for l in range(num_layers):
        I = input_size if l == 0 else hidden
        w((4 * hidden, I + hidden), 0.6 / np.sqrt(I + hidden))
        w((4 * hidden,), 0.2)
        w((hidden,), 0.1)
        w((hidden,), 0.1)
    w((out_channels, hidden), 1.0 / np.sqrt(hidden))
    w((out_channels,), 0.1)
    config = dict(input_size=input_size, hidden_size=hidden, num_layers=num_layers)
    if input_size != 1:
        config["in_channels"] = input_size
    if out_channels != 1:
        config["out_channels"] = out_channels
    model = dict(version="0.5.4", architecture="LSTM", config=config,
                 metadata=dict(name=name, note="synthetic test fixture (seeded random weights)"), weights=weights, sample_rate=48000)
    with open(os.path.join(HERE, "models", name + ".nam"), "w") as f:
        json.dump(model, f)
    return len(weights)


LSTMS = {
    # 18 hidden units = 5 unit tiles with a partial one, two layers: exercises the layer-to-layer k-steps
    "synth_lstm_h18x2": dict(num_layers=2, input_size=1, hidden=18, out_channels=1, seed=31),
    # 2 inputs / 3 outputs, one layer of 8
    "synth_lstm_io": dict(num_layers=1, input_size=2, hidden=8, out_channels=3, seed=32),
    # two layers of 10 (3 unit tiles): the register-resident kernel's layer-to-layer path
    "synth_lstm_h10x2": dict(num_layers=2, input_size=1, hidden=10, out_channels=1, seed=33),
    # small cells for the gate-row kernel (hidden <= 4): two layers of 4 (full 16-lane rows, layer-to-layer broadcasts),
    # and one layer of 2 with 2 inputs / 3 outputs (padding units, one output channel per lane)
    "synth_lstm_h4x2": dict(num_layers=2, input_size=1, hidden=4, out_channels=1, seed=34),
    "synth_lstm_h2io": dict(num_layers=1, input_size=2, hidden=2, out_channels=3, seed=35),
    # the two-rows-per-lane kernel at its limits: 32 units (every lane holds rows, 64 broadcasts per step) and two
    # layers of 24 with 2 inputs / 2 outputs
    "synth_lstm_h32": dict(num_layers=1, input_size=1, hidden=32, out_channels=1, seed=36),
    "synth_lstm_h24x2io": dict(num_layers=2, input_size=2, hidden=24, out_channels=2, seed=37),
}


def build_ktap(name, C, ksizes, dils, act, head_k, head_dil, head_bias, seed):
    """Single layer array with per-layer kernel sizes and a head rechannel with taps (the A2 shape family,
    NAM/wavenet/model.cpp:399-400, 547-548): reaches nam_kt_mfma_kernel (any kernel size <= 16, chunks of 6 taps).
    Stream order: rechannel [C][1]; per layer conv [C][C][K], bias [C], mixin [C][1], layer1x1 [C][C], bias [C];
    head rechannel [1][C][K_h] (+ bias [1]); head_scale."""
    rng = np.random.default_rng(seed)
    weights = []

    def w(shape, fan_in):
        v = rng.standard_normal(shape).astype(np.float32) * np.float32(0.9 / np.sqrt(fan_in))
        weights.extend(v.reshape(-1).tolist())

    head = dict(out_channels=1, kernel_size=head_k, bias=head_bias)
    if head_dil != 1:
        head["head_dilation"] = head_dil
    layer = dict(input_size=1, condition_size=1, head=head, channels=C, kernel_sizes=ksizes, dilations=dils, activation=act,
                 gated=False)
    w((C, 1), 1.0)
    for K in ksizes:
        w((C, C, K), C * K)
        w((C,), 4.0)
        w((C, 1), 1.0)
        w((C, C), C)
        w((C,), 4.0)
    w((1, C, head_k), C * head_k * len(ksizes))
    if head_bias:
        w((1,), 4.0)
    weights.append(0.05)
    model = dict(version="0.5.4", architecture="WaveNet", config=dict(layers=[layer], head=None, head_scale=0.05),
                 metadata=dict(name=name, note="synthetic test fixture (seeded random weights)"), weights=weights, sample_rate=48000)
    with open(os.path.join(HERE, "models", name + ".nam"), "w") as f:
        json.dump(model, f)
    return len(weights)


KTAP = {
    # half layout (C = 8): kernel sizes 1..16 -> 1, 2 and 3 chunks per layer, lookbacks inside / across / far beyond a
    # block, LeakyReLU (compile-time activation), head rechannel with 5 taps at dilation 2
    "synth_kt_c8": dict(C=8, ksizes=[6, 2, 7, 1, 13, 3, 16, 6, 4, 9, 6, 5], dils=[1, 3, 7, 2, 5, 17, 1, 41, 101, 13, 239, 2],
                        act=dict(type="LeakyReLU", negative_slope=0.02), head_k=5, head_dil=2, head_bias=True, seed=41),
    # full layout (C = 16), Tanh (Fasttanh when enabled), 16-tap head without bias
    "synth_kt_c16": dict(C=16, ksizes=[3, 6, 8, 2, 15, 1, 6, 12, 4, 7], dils=[1, 2, 64, 128, 1, 9, 300, 3, 33, 11], act="Tanh",
                         head_k=16, head_dil=1, head_bias=False, seed=42),
    # 12 channels (full layout, partial quad), two chunks per layer, ReLU (run-time activation dispatch), 1-tap head
    "synth_kt_c12": dict(C=12, ksizes=[7] * 9, dils=[1, 2, 4, 8, 16, 32, 64, 128, 256], act="ReLU", head_k=1, head_dil=1,
                         head_bias=True, seed=43),
    # 4 channels, 2 taps per layer, Sigmoid
    "synth_kt_c4": dict(C=4, ksizes=[2] * 16, dils=[1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 1, 2, 4, 8, 16], act="Sigmoid",
                        head_k=3, head_dil=1, head_bias=True, seed=44),
}


if __name__ == "__main__":
    for name, spec in KTAP.items():
        print(name, build_ktap(name, **spec), "weights")
    for name, spec in LSTMS.items():
        print(name, build_lstm(name, **spec), "weights")
    for name, spec in SPECS.items():
        print(name, build(name, **spec), "weights")
    for name, spec in GENERAL.items():
        print(name, build_general(name, **spec), "weights")
