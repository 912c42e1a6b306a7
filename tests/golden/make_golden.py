"""Regenerates tests/golden/outputs.npz: oracle outputs for every fixture model.

The reference repository stores no golden output vectors (its model-level tests only assert
isfinite), so these are produced by the CPU oracle (oracle/nam_oracle.{c,py}) — itself pinned bit for bit against
the reference's own sources built here on an Eigen stand-in (oracle/Makefile.ref, tests/test_reference_build.py),
against the reference's primitive KATs (tests/test_oracle_kat.py) and an independent PyTorch implementation
(tests/test_oracle_torch_crosscheck.py). Protocol per model: Reset(48000, 64) with prewarm, then
10 x 64 frames of the two-tone test signal of tools/test/test_a2_fast.cpp:118-128.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nam_oracle  # noqa: E402
from signals import two_tone  # noqa: E402

MODELS = ["wavenet", "wavenet_a1_standard", "lstm", "wavenet_a2_max", "slimmable_wavenet", "wavenet_condition_dsp",
          "synth_a1_13", "synth_a1_c12", "synth_a1_c8", "synth_a1_mixed",  # synth_*: make_synthetic_models.py
          "synth_a1_nano",  # 4 -> 2 channels: the register-resident kernel's plain-layer runs
          "synth_a1_lite", "synth_a1_c14", "synth_a1_feather",  # channel counts that are not multiples of 4 (zero-padded for the MFMA kernel)
          "A2", "slimmable_container",  # SlimmableContainer files: the default (last) submodel
          "synth_lstm_h4x2",  # a small LSTM cell (gate-row kernel)
          "synth_posthead",  # post-stack head, two output channels (mono input)
          "synth_leakyhardtanh",  # LeakyHardtanh, object / string forms, driven past both knees
          "synth_kt_c8", "synth_kt_c16", "synth_kt_c12", "synth_kt_c4"]  # per-layer kernel sizes, head rechannel with taps
# container submodels below the top one: (file, SetSlimmableSize value)
CONTAINER_RATIOS = [("A2", 0.2), ("slimmable_container", 0.1), ("slimmable_container", 0.5)]


def main():
    x = two_tone(640)
    out = {"input": x}
    for name in MODELS:
        for ft in (0, 1):
            m = nam_oracle.get_dsp(os.path.join(HERE, "models", name + ".nam"), fast_tanh=bool(ft))
            m.Reset(48000.0, 64)
            out[f"{name}__ft{ft}"] = m.process_stream(x, 64).astype(np.float32)
    m = nam_oracle.get_dsp(os.path.join(HERE, "models", "slimmable_wavenet.nam"))
    for ratio, tag in ((0.0, "w1"), (0.34, "w2")):
        m.SetSlimmableSize(ratio)
        m.Reset(48000.0, 64)
        out[f"slimmable_wavenet__{tag}"] = m.process_stream(x, 64).astype(np.float32)
    for name, ratio in CONTAINER_RATIOS:
        m = nam_oracle.get_dsp(os.path.join(HERE, "models", name + ".nam"), fast_tanh=True)
        m.SetSlimmableSize(ratio)
        m.Reset(48000.0, 64)
        out[f"{name}__c{ratio}"] = m.process_stream(x, 64).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "outputs.npz"), **out)
    print("wrote", os.path.join(HERE, "outputs.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
