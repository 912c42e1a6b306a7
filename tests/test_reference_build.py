"""Pins the oracle to the REFERENCE ITSELF: oracle/_ref/libnam_ref.so is the reference's own, unmodified C++
sources (NAM/{activations,conv1d,dsp,get_dsp,lstm,nam_file,ring_buffer,util,container}.cpp,
NAM/wavenet/{model,slimmable}.cpp) compiled where they lie against the self-written Eigen stand-in of
oracle/eigen_shim (the real Eigen is absent here). The oracle restatement must reproduce it exactly — the two
sum every dot product in the same order and both are built without FMA contraction — on every fixture, both tanh
modes, every slimmable width / container submodel, for ragged buffer sizes, and the committed golden vectors must
be the reference's outputs. Skipped where neither /root/reference nor a prebuilt library exists."""
import glob
import os
import sys

import numpy as np
import pytest

from conftest import MODELS, ROOT, model_path

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nam_ref  # noqa: E402

pytestmark = pytest.mark.skipif(not nam_ref.available(), reason="reference sources / prebuilt oracle/_ref not available")

ALL = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(MODELS, "*.nam")))
G = np.load(os.path.join(ROOT, "tests", "golden", "outputs.npz"))


def _signal(in_ch, n, seed=7):
    if in_ch == 1:
        return G["input"][:n] if n <= len(G["input"]) else np.resize(G["input"], n)
    return np.random.default_rng(seed).uniform(-0.5, 0.5, (in_ch, n)).astype(np.float32)


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("fast_tanh", [False, True])
def test_oracle_is_bit_exact_with_the_reference(oracle, name, fast_tanh):
    ref = nam_ref.get_dsp(model_path(name), fast_tanh)
    orc = oracle.get_dsp(model_path(name), fast_tanh=fast_tanh)
    assert (ref.NumInputChannels(), ref.NumOutputChannels()) == (orc.NumInputChannels(), orc.NumOutputChannels())
    assert ref.GetPrewarmSamples() == orc.GetPrewarmSamples()
    x = _signal(ref.NumInputChannels(), 640)
    for block, n in ((64, 640), (37, 500), (256, 640)):  # Reset(…, block) fixes the prewarm length, then ragged calls
        ref.Reset(48000.0, block)
        orc.Reset(48000.0, block)
        xs = x[..., :n]
        np.testing.assert_array_equal(ref.process_stream(xs, block), orc.process_stream(xs, block))


@pytest.mark.parametrize("name,ratios", [("slimmable_wavenet", (0.0, 0.34, 0.5, 0.67, 1.0)), ("A2", (0.0, 0.49, 0.5, 1.0)),
                                          ("slimmable_container", (0.0, 0.32, 0.33, 0.5, 0.66, 0.9))])
def test_slimmable_sizes_bit_exact(oracle, name, ratios):
    x = _signal(1, 400)
    for fast_tanh in (False, True):
        ref = nam_ref.get_dsp(model_path(name), fast_tanh)
        orc = oracle.get_dsp(model_path(name), fast_tanh=fast_tanh)
        ref.Reset(48000.0, 64)
        orc.Reset(48000.0, 64)
        for r in ratios:  # switching an initialised model: the newly selected size is Reset + prewarmed
            ref.SetSlimmableSize(r)
            orc.SetSlimmableSize(r)
            assert ref.GetPrewarmSamples() == orc.GetPrewarmSamples()
            np.testing.assert_array_equal(ref.process_stream(x, 64), orc.process_stream(x, 64))


@pytest.mark.parametrize("ratio", [0.0, 1.0])
def test_oracle_matches_the_reference_a2_fast_path(oracle, ratio):
    """The reference's default build runs A2-shaped files through wavenet/a2_fast.cpp (NAM_ENABLE_A2_FAST,
    CMakeLists.txt:58, dispatch model.cpp:1317) — libnam_ref_a2fast.so is that build. Its own A/B protocol against
    the generic WaveNet (tools/test/test_a2_fast.cpp:109-128,272-300: two-tone, 2,048 frames, blocks 64 and 256,
    5e-5) applied to fast path vs generic reference vs oracle, plus the prewarm-count guard (:307-325)."""
    from signals import two_tone
    x = two_tone(2048)
    for block in (64, 256):
        fast = nam_ref.get_dsp(model_path("A2"), a2_fast=True)
        gen = nam_ref.get_dsp(model_path("A2"))
        orc = oracle.get_dsp(model_path("A2"))
        outs = []
        for d in (fast, gen, orc):
            d.Reset(48000.0, block)
            d.SetSlimmableSize(ratio)
            outs.append(d.process_stream(x, block))
        assert fast.GetPrewarmSamples() == gen.GetPrewarmSamples() == orc.GetPrewarmSamples()
        np.testing.assert_array_equal(outs[1], outs[2])  # generic reference == oracle, bit for bit
        assert float(np.max(np.abs(outs[0] - outs[2]))) <= 5e-5  # fast path within the reference's own bound
        assert float(np.max(np.abs(outs[0]))) > 0.05


@pytest.mark.parametrize("name,lut", [("wavenet", ("Tanh", -5.0, 5.0, 1024)), ("wavenet_a1_standard", ("Tanh", -4.0, 4.0, 4096)),
                                      ("wavenet_a2_max", ("Sigmoid", -8.0, 8.0, 1024)), ("wavenet_a2_max", ("SiLU", -6.0, 6.0, 333))])
def test_lookup_table_activations_bit_exact(oracle, name, lut):
    """Activation::enable_lut(name, min, max, n) before get_dsp (activations.cpp:189-212, FastLUTActivation
    activations.h:371-422; the reference's own check is tools/test/test_fast_lut.cpp): oracle == reference, and the
    table actually changes the output."""
    x = _signal(1, 640)
    ref = nam_ref.get_dsp(model_path(name), lut=lut)
    orc = oracle.get_dsp(model_path(name), luts={lut[0]: lut[1:]})
    plain = oracle.get_dsp(model_path(name))
    for d in (ref, orc, plain):
        d.Reset(48000.0, 64)
    y = orc.process_stream(x, 64)
    np.testing.assert_array_equal(ref.process_stream(x, 64), y)
    assert float(np.max(np.abs(y - plain.process_stream(x, 64)))) > 0.0


@pytest.mark.parametrize("key", [k for k in G.files if "__ft" in k])
def test_committed_goldens_are_the_reference_outputs(key):
    name, ft = key.split("__ft")
    ref = nam_ref.get_dsp(model_path(name), bool(int(ft)))
    ref.Reset(48000.0, 64)
    y = ref.process_stream(G["input"], 64)
    # exact where no libm transcendental is involved; a couple of ulp of slack otherwise (golden files are
    # generated on one glibc, replayed on another)
    np.testing.assert_allclose(y, G[key], rtol=0, atol=2e-6 * max(1.0, float(np.max(np.abs(G[key])))))


def test_reference_error_messages_match_the_loader(nam_lib, tmp_path):
    """The product's loader reports what the reference's get_dsp reports for the same bad files."""
    import json
    with open(model_path("wavenet")) as f:
        good = json.load(f)
    cases = {}
    bad = json.loads(json.dumps(good))
    bad["weights"] = bad["weights"][:-3]
    cases["short"] = bad
    bad = json.loads(json.dumps(good))
    bad["weights"] = bad["weights"] + [0.0, 0.0]
    cases["long"] = bad
    bad = json.loads(json.dumps(good))
    bad["architecture"] = "Transformer"
    cases["arch"] = bad
    bad = json.loads(json.dumps(good))
    bad["version"] = "0.4.0"
    cases["version"] = bad
    for tag, j in cases.items():
        p = str(tmp_path / f"{tag}.nam")
        with open(p, "w") as f:
            json.dump(j, f)
        with pytest.raises(RuntimeError) as ref_err:
            nam_ref.get_dsp(p)
        with pytest.raises(Exception) as our_err:
            nam_lib.get_dsp(p)
        assert str(ref_err.value) in str(our_err.value), (tag, str(ref_err.value), str(our_err.value))


@pytest.mark.parametrize("seed", range(50))
def test_featured_models_oracle_bit_exact_with_the_reference(oracle, nam_lib, tmp_path, seed):
    """The seeded feature-rich models of tests/test_gpu_breadth.py (per-layer gating / blending, FiLM subsets, grouped
    convs, head1x1, bottlenecks, nested condition_dsp — the cases of the reference's own feature tests,
    tools/test/test_wavenet_configurable_gating.cpp:86-264, test_film.cpp:26-480, test_wavenet/test_head1x1.cpp,
    test_wavenet/test_condition_processing.cpp): the reference loads every one (its weight-count check is the pin of
    the generator), the oracle reproduces it bit for bit in both tanh modes, and the product's loader accepts it."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_synthetic_models as msm
    path = str(tmp_path / f"featured_{seed}.nam")
    msm.write_featured(path, 7000 + seed, wr_shapes=bool(seed % 2), post_head=seed >= 40)
    model = nam_lib.get_dsp(path)
    for fast_tanh in (False, True):
        ref = nam_ref.get_dsp(path, fast_tanh)
        orc = oracle.get_dsp(path, fast_tanh=fast_tanh)
        assert (ref.NumInputChannels(), ref.NumOutputChannels()) == (orc.NumInputChannels(), orc.NumOutputChannels()) == (
            model.NumInputChannels(), model.NumOutputChannels())
        assert ref.GetPrewarmSamples() == orc.GetPrewarmSamples() == model.GetPrewarmSamples()
        x = _signal(ref.NumInputChannels(), 300, seed=seed)
        for block in (64, 37):
            ref.Reset(48000.0, block)
            orc.Reset(48000.0, block)
            np.testing.assert_array_equal(ref.process_stream(x, block), orc.process_stream(x, block))
