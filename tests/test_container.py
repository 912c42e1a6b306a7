"""SlimmableContainer (NAM/container.cpp): dispatch by max_value, construction checks, breakpoints — in the oracle
and in the product's loader (host side of libnam_hip.so; no GPU needed). The reference's own test of this
behaviour is tools/test/test_container.cpp; the files are its example_models/{A2,slimmable_container}.nam."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, model_path

G = np.load(os.path.join(ROOT, "tests", "golden", "outputs.npz"))


def _container(entries, sample_rate=48000):
    subs = []
    for max_value, name in entries:
        with open(model_path(name)) as f:
            subs.append({"max_value": max_value, "model": json.load(f)})
    return {"version": "0.7.0", "architecture": "SlimmableContainer", "config": {"submodels": subs}, "weights": [],
            "sample_rate": sample_rate}


@pytest.mark.parametrize("key", [k for k in G.files if "__c" in k])
def test_oracle_container_goldens(oracle, key):
    name, ratio = key.split("__c")
    m = oracle.get_dsp(model_path(name), fast_tanh=True)
    m.SetSlimmableSize(float(ratio))
    m.Reset(48000.0, 64)
    y = m.process_stream(G["input"], 64)
    np.testing.assert_allclose(y, G[key], rtol=0, atol=2e-6 * max(1.0, float(np.max(np.abs(G[key])))))


def test_oracle_dispatch_rule_and_equivalence_to_the_bare_submodel(oracle, tmp_path):
    """container.cpp:103-115: first submodel with val < max_value, else the last; the active submodel's output
    is exactly that model's own output (the container only forwards)."""
    m = oracle.get_dsp(model_path("slimmable_container"))
    assert m.GetSlimmableSizeBreakpoints() == [0.33, 0.66]
    assert [m.index_for(v) for v in (0.0, 0.329, 0.33, 0.5, 0.66, 0.99, 1.0, 7.0)] == [0, 0, 1, 1, 2, 2, 2, 2]
    with open(model_path("slimmable_container")) as f:
        j = json.load(f)
    x = G["input"]
    for val, idx in ((0.1, 0), (0.5, 1), (1.0, 2)):
        p = str(tmp_path / f"sub{idx}.nam")
        with open(p, "w") as f:
            json.dump(j["config"]["submodels"][idx]["model"], f)
        bare = oracle.get_dsp(p)
        bare.Reset(48000.0, 64)
        m2 = oracle.get_dsp(model_path("slimmable_container"))
        m2.SetSlimmableSize(val)
        m2.Reset(48000.0, 64)
        np.testing.assert_array_equal(m2.process_stream(x, 64), bare.process_stream(x, 64))
        assert m2.GetPrewarmSamples() == bare.GetPrewarmSamples()


BAD = [
    (lambda: _container([(0.5, "wavenet"), (0.5, "lstm")]), "sorted by ascending max_value"),
    (lambda: _container([(0.3, "wavenet"), (0.9, "lstm")]), "last submodel max_value must be >= 1.0"),
    (lambda: _container([(1.0, "wavenet")], sample_rate=44100), "sample rate mismatch"),
    (lambda: {**_container([(1.0, "wavenet")]), "config": {"submodels": []}}, "'submodels' must be a non-empty array"),
]


@pytest.mark.parametrize("case", range(len(BAD)))
def test_construction_errors_oracle_and_product(nam_lib, oracle, case):
    make, needle = BAD[case]
    j = make()
    with pytest.raises(RuntimeError, match=needle):
        oracle.load_nam_json(j)
    with pytest.raises(Exception, match=needle):
        nam_lib.get_dsp_json(json.dumps(j))


def test_product_loader_container_info(nam_lib):
    m = nam_lib.get_dsp(model_path("A2"))
    assert m.architecture == "SlimmableContainer" and m.info.is_slimmable == 1
    assert (m.NumInputChannels(), m.NumOutputChannels()) == (1, 1)
    assert m.GetPrewarmSamples() == 6347  # of the default (last) submodel: receptive field 6,346 + 1
    assert list(m.GetSlimmableSizeBreakpoints()) == [0.5]
    assert m.info.has_loudness == 1 and abs(m.info.expected_sample_rate - 48000.0) < 1e-9
    m = nam_lib.get_dsp(model_path("slimmable_container"))
    assert list(m.GetSlimmableSizeBreakpoints()) == [0.33, 0.66]
    assert m.GetPrewarmSamples() == 4093
