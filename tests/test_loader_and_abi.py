"""Host logic without a GPU: the product's .nam loader / plan compiler against the oracle's
independent (Python) reading of the same files, reference error behaviour, and the C ABI surface."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import MODELS, ROOT, model_path

ALL = ["wavenet", "wavenet_a1_standard", "lstm", "wavenet_a2_max", "slimmable_wavenet", "wavenet_condition_dsp"]
EXPECTED_WEIGHTS = {"wavenet": 131, "wavenet_a1_standard": 13802, "lstm": 70, "wavenet_a2_max": 818,
                    "slimmable_wavenet": 457, "wavenet_condition_dsp": 147}


@pytest.mark.parametrize("name", ALL)
def test_loader_agrees_with_oracle(nam_lib, oracle, name):
    m = nam_lib.get_dsp(model_path(name))
    o = oracle.get_dsp(model_path(name))
    assert m.num_weights == EXPECTED_WEIGHTS[name]
    assert m.NumInputChannels() == o.NumInputChannels() and m.NumOutputChannels() == o.NumOutputChannels()
    assert m.GetPrewarmSamples() == o.GetPrewarmSamples()
    assert m.GetExpectedSampleRate() == o.expected_sample_rate
    assert m.HasLoudness() == (o.loudness is not None)
    if m.HasLoudness():
        assert m.GetLoudness() == pytest.approx(o.loudness)


def test_a1_standard_has_no_sample_rate_and_fast_kernels(nam_lib):
    m = nam_lib.get_dsp(model_path("wavenet_a1_standard"))
    assert m.GetExpectedSampleRate() == -1.0 and not m.HasLoudness()  # get_dsp.cpp:275-281
    assert m.GetPrewarmSamples() == 4093  # model.cpp:653-658
    assert m.info.has_a1_kernel == 15  # VALU + MFMA + interleaved-frame MFMA (+ its compile-time-topology form)
    # write-position table (64 words) + one ring of (K-1)*d + 64 frames per layer, rounded up to 64 floats
    floats = 64 + sum(c * (2 * d + 64) for c in (16, 8) for d in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512))
    assert m.info.state_bytes_per_stream == 4 * ((floats + 63) // 64 * 64)
    with pytest.raises(RuntimeError):
        m.GetLoudness()  # dsp.cpp:121-128


@pytest.mark.parametrize("name,bits", [
    ("wavenet_a1_standard", 15), ("A2", 3), ("synth_kt_c8", 3), ("synth_kt_c16", 3), ("synth_kt_c12", 3),
    ("synth_kt_c4", 19),  # (4 channels, 2 taps per layer: small enough for nam_wn_reg_kernel compiled for its shapes; AUTO keeps the matrix cores)
    ("synth_a1_mixed", 3),  # kernel size 3 everywhere, several arrays: the wave-specialised MFMA kernel (the interleaved-frame ones: official sizes only)
    ("synth_a1_nano", 17),  # 4 -> 2 channels: VALU kernel, and nam_wn_reg_kernel's plain-layer runs (68 KB of LDS rings)
    ("synth_a1_lite", 15), ("synth_a1_c14", 3),
    ("synth_a1_feather_relu", 31), ("synth_a1_feather", 31),  # (8 -> 4 channels also fit nam_wn_reg_kernel: 129 KB of LDS rings; AUTO keeps the matrix cores)  # 6 / 14 / 10 channels: zero-padded to a multiple of 4 for them
    ("slimmable_wavenet", 17),  # 3 channels: VALU kernel, and (dilations up to 512 in LDS-resident rings) nam_wn_reg_kernel
    # FiLMs / gating / nested condition_dsp / multi-channel: the register-resident kernel (bit 4) where every layer is
    # one of its instantiated shapes, else the op interpreter alone (a post-stack head)
    ("wavenet_a2_max", 16), ("wavenet_condition_dsp", 16), ("synth_multich", 16), ("synth_leakyhardtanh", 16),
    ("synth_posthead", 16), ("wavenet", 17),
    ("lstm", 0)])
def test_kernel_eligibility_reported_by_the_plan_compiler(nam_lib, name, bits):
    """has_a1_kernel: bit 0 = the VALU A1 kernel, bit 1 = one of the MFMA kernels (plan_a1.cpp: build_a1 / build_a1_ws /
    build_a1_kt), bit 2 = the interleaved-frame MFMA kernel (build_a1_il), bit 3 = its compile-time-topology
    form for the official sizes (plan.h: namespace p2), bit 4 = nam_wn_reg_kernel (plan_wr.cpp: build_wr). Decided on the host at load time, so it is checkable without a GPU."""
    assert nam_lib.get_dsp(model_path(name)).info.has_a1_kernel == bits


def test_the_a2_pipeline_kernel_is_offered_to_the_a2_topology_only(nam_lib):
    """nam_kq_kernel is compiled for ONE topology (csrc/kp_table.h: the A2 stack of the reference's fused path,
    wavenet/a2_fast.cpp): plan_a1.cpp: build_a1_kp checks a model against the table layer by layer — kernel sizes, dilations,
    ring geometry, chunk and tile offsets — and only A2.nam's 8-channel submodel passes; other K-tap models keep the
    descriptor-driven kernel (kt_mfma=1, kp=0); A2-Lite (3 channels) runs on nam_wn_reg_kernel."""
    d = nam_lib.get_dsp(model_path("A2")).describe()
    lite, full = d.split(" | plan 1:")
    assert " kt_mfma=1 kp=1 " in full and " kp=0 " in lite and " wn_reg=1" in lite
    for other in ("synth_kt_c8", "synth_kt_c16", "wavenet_a1_standard"):
        assert " kp=0 " in nam_lib.get_dsp(model_path(other)).describe(), other
    assert " kt_mfma=1 " in nam_lib.get_dsp(model_path("synth_kt_c8")).describe()


def test_missing_file_is_validation_error(nam_lib):
    with pytest.raises(nam_lib.NamFileValidationError):
        nam_lib.get_dsp(os.path.join(MODELS, "does_not_exist.nam"))


def _variant(name, mutate):
    j = json.load(open(model_path(name)))
    mutate(j)
    return json.dumps(j)


def test_reference_error_behaviour(nam_lib, tmp_path):
    nam = nam_lib
    # missing required key -> NamFileValidationError (nam_file.cpp:31-37)
    p = tmp_path / "nokey.nam"
    p.write_text(_variant("wavenet", lambda j: j.pop("weights")))
    with pytest.raises(nam.NamFileValidationError):
        nam.get_dsp(str(p))
    p = tmp_path / "garbage.nam"
    p.write_text("{ not json")
    with pytest.raises(nam.NamFileValidationError):
        nam.get_dsp(str(p))
    # unsupported versions (get_dsp.cpp:18-39): too old, minor too new, non-semver
    for v in ("0.4.9", "0.8.0", "1.0.0", "0.5"):
        with pytest.raises(nam.NamHipError, match="unsupported version"):
            nam.get_dsp_json(_variant("wavenet", lambda j, v=v: j.__setitem__("version", v)))
    nam.get_dsp_json(_variant("wavenet", lambda j: j.__setitem__("version", "0.7.3")))  # partial support: loads
    # weight count mismatch (model.cpp:671-682)
    with pytest.raises(nam.NamHipError, match="Weight mismatch"):
        nam.get_dsp_json(_variant("wavenet", lambda j: j["weights"].append(0.0)))
    with pytest.raises(nam.NamHipError, match="Weight mismatch"):
        nam.get_dsp_json(_variant("wavenet", lambda j: j["weights"].pop()))
    # unknown architecture (model_config.h:84-87)
    with pytest.raises(nam.NamHipError, match="No config parser registered"):
        nam.get_dsp_json(_variant("wavenet", lambda j: j.__setitem__("architecture", "ConvNet")))
    # kernel_size and kernel_sizes together (model.cpp:1004-1008)
    def both(j):
        j["config"]["layers"][0]["kernel_sizes"] = [3, 3]
    with pytest.raises(nam.NamHipError, match="only one of kernel_size"):
        nam.get_dsp_json(_variant("wavenet", both))
    with pytest.raises(nam.NamHipError, match="Unknown activation"):
        nam.get_dsp_json(_variant("wavenet", lambda j: j["config"]["layers"][0].__setitem__("activation", "Nope")))


def test_legacy_and_new_schema_forms_load_identically(nam_lib, oracle):
    """`gated: false` == gating_mode none; kernel_size == kernel_sizes; head_size/head_bias == nested head."""
    def modern(j):
        for lc in j["config"]["layers"]:
            n = len(lc["dilations"])
            lc["kernel_sizes"] = [lc.pop("kernel_size")] * n
            lc["gating_mode"] = "none"
            lc.pop("gated")
            lc["head"] = {"out_channels": lc.pop("head_size"), "kernel_size": 1, "bias": lc.pop("head_bias")}
            lc["activation"] = {"type": lc["activation"]}
    a = nam_lib.get_dsp(model_path("wavenet"))
    b = nam_lib.get_dsp_json(_variant("wavenet", modern))
    assert (a.num_weights, a.GetPrewarmSamples(), a.info.state_bytes_per_stream) == (
        b.num_weights, b.GetPrewarmSamples(), b.info.state_bytes_per_stream)
    x = (0.2 * np.sin(0.05 * np.arange(256))).astype(np.float32)
    oa = oracle.get_dsp(model_path("wavenet"))
    ob = oracle.load_nam_json(json.loads(_variant("wavenet", modern)))
    oa.Reset(48000, 64)
    ob.Reset(48000, 64)
    np.testing.assert_array_equal(oa.process_stream(x, 64), ob.process_stream(x, 64))


def test_slimmable_breakpoints_and_ratio_mapping(nam_lib, oracle):
    m = nam_lib.get_dsp(model_path("slimmable_wavenet"))
    assert m.is_slimmable and m.GetSlimmableSizeBreakpoints() == pytest.approx([1 / 3, 2 / 3])
    o = oracle.get_dsp(model_path("slimmable_wavenet"))
    assert o.GetSlimmableSizeBreakpoints() == pytest.approx([1 / 3, 2 / 3])
    # idx = min(floor(ratio * len), len - 1) (slimmable.cpp:102-106)
    assert [o.channels_for(r)[0] for r in (0.0, 0.33, 0.34, 0.66, 0.67, 1.0)] == [1, 1, 2, 2, 3, 3]
    assert not nam_lib.get_dsp(model_path("wavenet")).is_slimmable


def test_library_exports_exactly_the_header_abi(nam_lib):
    """Every symbol include/nam_hip.h declares is exported by the built library, and nothing else is."""
    header = open(os.path.join(ROOT, "include", "nam_hip.h")).read()
    declared = sorted(set(re.findall(r"NAM_HIP_API\s+[\w\s\*]*?\b(nam_hip_\w+)\s*\(", header)))
    out = subprocess.check_output(["nm", "-D", "--defined-only", nam_lib.lib_path()], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and "nam_hip_" in l)
    assert declared == exported == sorted(nam_lib.ABI_SYMBOLS)
    assert nam_lib.load_library().nam_hip_version().decode().endswith("gfx950")


def test_library_carries_gfx950_code_objects(nam_lib):
    """The shared library embeds device code for gfx950 only (no CUDA / multi-arch fat binary)."""
    data = open(nam_lib.lib_path(), "rb").read()
    assert b"gfx950" in data
    assert b"gfx942" not in data and b"sm_" not in data


def test_no_cpu_fallback_without_library(monkeypatch, nam_lib):
    import neuralampmodelercore_amd as nam
    monkeypatch.setattr(nam, "_lib", None)
    monkeypatch.setattr(nam, "lib_path", lambda: os.path.join(ROOT, "no_such_dir", "libnam_hip.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        nam.get_dsp(model_path("wavenet"))


def test_lookup_table_load_option(nam_lib):
    """Activation::enable_lut as a load option (activations.cpp:189-212): accepted names, the reference's message for
    any other, and a model with a table is not handed to the register-resident kernels."""
    nam = nam_lib
    m = nam.get_dsp(model_path("wavenet_a1_standard"), luts={"Tanh": (-5.0, 5.0, 1024)})
    assert m.info.has_a1_kernel == 0  # tables are interpolated by the op-program interpreter only
    assert nam.get_dsp(model_path("wavenet_a1_standard")).info.has_a1_kernel == 15
    with pytest.raises(nam.NamHipError) as e:
        nam.get_dsp(model_path("wavenet"), luts={"ReLU": (-1.0, 1.0, 16)})
    assert "Tried to enable LUT for a function other than Tanh, Sigmoid, or SiLU" in str(e.value)
    with pytest.raises(nam.NamHipError):
        nam.get_dsp(model_path("wavenet"), luts={"Tanh": (1.0, -1.0, 16)})
    with open(model_path("wavenet")) as f:
        assert nam.get_dsp_json(f.read(), luts={"Sigmoid": (-8.0, 8.0, 64)}).NumOutputChannels() == 1


@pytest.mark.parametrize("field,value", [("dilations", [1, -2]), ("dilations", [1, 0]), ("dilations", [1, 1 << 28]),
                                          ("dilations", [1 << 25, 1 << 25]), ("kernel_size", 0)])
def test_corrupt_geometry_is_a_load_error(nam_lib, tmp_path, field, value):
    """A crafted .nam (negative / zero / huge dilation, zero kernel size) must fail at load instead of overflowing the
    32-bit state offsets the kernels use (the reference would run out of memory)."""
    import json
    with open(model_path("wavenet")) as f:
        j = json.load(f)
    j["config"]["layers"][0][field] = value
    p = str(tmp_path / "bad.nam")
    with open(p, "w") as f:
        json.dump(j, f)
    with pytest.raises(nam_lib.NamHipError):
        nam_lib.get_dsp(p)


def test_register_resident_kernel_state_is_an_image_of_its_lds_rings(nam_lib):
    """nam_wn_reg_kernel's per-stream state: 64 write positions, then one ring per layer of exactly
    (K - 1) * dilation + 64 frames, stored in groups of up to four channels and padded to four floats; rounded up to 64
    floats and never smaller than the other kernels' layout of the same model. Decided by the plan compiler: checkable
    without a GPU."""
    def wr_floats(layers):  # [(channels, kernel, dilation)]
        return (64 + sum((c * ((k - 1) * d + 64) + 3) // 4 * 4 for c, k, d in layers) + 63) // 64 * 64

    D = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
    m = nam_lib.get_dsp(model_path("slimmable_wavenet"))  # full width: 3 channels
    assert m.info.state_bytes_per_stream == 4 * wr_floats([(3, 3, d) for d in D]) == 32512
    m = nam_lib.get_dsp(model_path("synth_a1_nano"))  # 4 -> 2 channels: 68 KB of rings, LDS-resident
    assert m.info.state_bytes_per_stream == 4 * wr_floats([(4, 3, d) for d in D] + [(2, 3, d) for d in D])
    # wavenet_a2_max: nested condition net (3 channels K = 2 x 2; 4 channels K = 3, dilations 1, 3, 5), main array (4
    # channels, K = 4, dilations 1, 2); the op program's rings of the same model are the larger layout here
    m = nam_lib.get_dsp(model_path("wavenet_a2_max"))
    assert m.info.state_bytes_per_stream >= 4 * wr_floats([(3, 2, 1), (3, 2, 2), (4, 3, 1), (4, 3, 3), (4, 3, 5), (4, 4, 1), (4, 4, 2)])


def test_per_model_compile_cache_directory_is_private(nam_lib, tmp_path, monkeypatch):
    """csrc/wr_jit.cpp loads what it finds in its cache onto the GPU: a cache directory someone else could write into (group-
    or world-writable, a symbolic link) is refused — the model then says why it is not on its compiled shapes —, a private one
    is used, and a path with a quote in it reaches the compiler as one argument (no shell)."""
    import stat
    sys_path_golden = os.path.join(ROOT, "tests", "golden")
    import sys
    if sys_path_golden not in sys.path:
        sys.path.insert(0, sys_path_golden)
    import make_synthetic_models as msm
    nam = nam_lib
    p = str(tmp_path / "featured.nam")
    msm.write_featured(p, 7001, wr_shapes=True, post_head=False)  # (seed 7001: shapes outside the ahead-of-time tables)
    open_dir = tmp_path / "open"
    open_dir.mkdir()
    os.chmod(open_dir, 0o777)
    monkeypatch.setenv("NAM_HIP_JIT_CACHE", str(open_dir))
    m = nam.get_dsp(p, fast_tanh=False)
    # loud: nam_hip_model_info bit 5 and the description say that the model runs off its compiled shapes, and why
    assert m.info.has_a1_kernel & 32 and "only this user can write" in m.jit_failed(), m.describe()
    assert not list(open_dir.iterdir())
    link = tmp_path / "link"
    private = tmp_path / "it's private"
    private.mkdir(mode=0o700)
    link.symlink_to(private)
    monkeypatch.setenv("NAM_HIP_JIT_CACHE", str(link))
    m = nam.get_dsp(p, fast_tanh=False)
    assert m.info.has_a1_kernel & 32, "a symbolic link is not a cache directory"
    monkeypatch.setenv("NAM_HIP_JIT_CACHE", str(private))
    m = nam.get_dsp(p, fast_tanh=False)
    assert m.info.has_a1_kernel & 16 and not (m.info.has_a1_kernel & 32) and m.jit_failed() == "", m.describe()
    objs = [f for f in private.iterdir() if f.name.endswith(".hsaco")]
    assert len(objs) == 1 and stat.S_IMODE(objs[0].stat().st_mode) & 0o022 == 0
    assert [f.name for f in private.iterdir()] == [objs[0].name]  # no header / log / temporary file left behind


def test_load_options_struct_size_guards_the_version_gate(nam_lib):
    """nam_hip_load_options grew once (library 0.1: 16 bytes). A caller built against the old header passes a struct that
    ends behind `luts`: whatever lies there must not switch the built-in version gate off — the library reads
    version_checked_by_caller only when struct_size covers it (include/nam_hip.h)."""
    import ctypes
    nam = nam_lib
    L = nam.load_library()
    assert "0.2" in L.nam_hip_version().decode()
    doc = json.load(open(model_path("wavenet")))
    doc["version"] = "0.9.0"  # beyond the built-in gate (minor <= 0.7)
    text = json.dumps(doc).encode()
    for struct_size, accepted in ((0, False), (16, False), (ctypes.sizeof(nam._LoadOptions), True)):
        opts = nam._LoadOptions(0, 0, None, 1, struct_size)  # version_checked_by_caller = 1
        h = ctypes.c_void_p()
        rc = L.nam_hip_model_load_ex(None, text, ctypes.byref(opts), ctypes.byref(h))
        assert (rc == 0) == accepted, (struct_size, rc, L.nam_hip_last_error())
        if rc == 0:
            L.nam_hip_model_free(h)


def test_config_and_metadata_text_round_trips_non_finite_numbers(nam_lib):
    """json_min.h writes what it reads: NaN / Infinity / -Infinity (no JSON form exists; Python's spellings), the sign of zero,
    integers only inside the range where the test is defined — nam::dspData's config / metadata text goes back through
    get_dsp(dspData&) (NAM/get_dsp.h:91), non-finite weights included."""
    nam = nam_lib
    doc = json.load(open(model_path("wavenet")))
    doc["metadata"] = {"loudness": -20.5, "gain": float("nan"), "hi": float("inf"), "lo": float("-inf"), "zero": -0.0,
                       "big": 1e300, "huge_int": 2 ** 70, "n": 3}
    m = nam.get_dsp_json(json.dumps(doc), fast_tanh=False)
    d = m.dsp_data()
    meta = json.loads(d["metadata"])
    assert np.isnan(meta["gain"]) and meta["hi"] == float("inf") and meta["lo"] == float("-inf")
    assert meta["zero"] == 0.0 and np.signbit(meta["zero"]) and meta["big"] == 1e300 and meta["huge_int"] == float(2 ** 70) and meta["n"] == 3
    conf = dict(d)  # (config / metadata stay JSON text, as nam::dspData carries them across the boundary)
    m2 = nam.get_dsp_data(conf, fast_tanh=False)
    assert json.loads(m2.dsp_data()["metadata"]).keys() == meta.keys()
    w = d["weights"].copy()
    w[3] = np.inf
    conf["weights"] = w
    m3 = nam.get_dsp_data(conf, fast_tanh=False)  # a non-finite weight travels as the spelling the parser reads
    assert np.isinf(m3.dsp_data()["weights"][3])


def test_ticket_entry_points_refuse_bad_arguments(nam_lib):
    """nam_hip_batch_submit_f32 / nam_hip_batch_wait_f32 without a batch (no device needed): an error code and a message, no crash"""
    import ctypes
    L = nam_lib.load_library()
    t = ctypes.c_int64(-1)
    x = (ctypes.c_float * 64)()
    assert L.nam_hip_batch_submit_f32(None, x, 64, ctypes.byref(t)) == nam_lib.ERR_INVALID_ARGUMENT
    assert b"nam_hip_batch_submit_f32" in L.nam_hip_last_error()
    assert L.nam_hip_batch_wait_f32(None, 0, x) == nam_lib.ERR_INVALID_ARGUMENT
    assert b"nam_hip_batch_wait_f32" in L.nam_hip_last_error()
    xd = (ctypes.c_double * 64)()
    assert L.nam_hip_batch_submit_f64(None, xd, 64, ctypes.byref(t)) == nam_lib.ERR_INVALID_ARGUMENT
    assert L.nam_hip_batch_wait_f64(None, 0, xd) == nam_lib.ERR_INVALID_ARGUMENT
    assert b"nam_hip_batch_wait_f64" in L.nam_hip_last_error()
    assert nam_lib.Batch.PIPE_SLOTS == int(re.search(r"#define NAM_HIP_PIPE_SLOTS (\d+)", open(os.path.join(ROOT, "include", "nam_hip.h")).read()).group(1))


def test_per_model_compile_survives_a_compiler_crash(nam_lib, tmp_path, monkeypatch):
    """ROCm 7.2's "Rewrite AGPR-Copy-MFMA" pass crashes on some register-hungry layer shapes when matrix-instruction results are
    forced into vector registers (found by tools/fuzz_models.py 160 7707, model 123: a gated 16-row layer with every FiLM on a
    4-value condition — the FiLMs run on v_mfma_f32_4x4x1 since round 6). csrc/wr_jit.cpp compiles once more without the flag:
    the model still gets its compiled shapes instead of the slow fallback, and nothing of the failed attempt stays behind."""
    import sys
    sys_path_golden = os.path.join(ROOT, "tests", "golden")
    if sys_path_golden not in sys.path:
        sys.path.insert(0, sys_path_golden)
    import make_synthetic_models as msm
    nam = nam_lib
    p = str(tmp_path / "featured_crash.nam")
    msm.write_featured(p, 752428520, wr_shapes=False, post_head=True)
    cache = tmp_path / "cache"
    cache.mkdir(mode=0o700)
    monkeypatch.setenv("NAM_HIP_JIT_CACHE", str(cache))
    m = nam.get_dsp(p, fast_tanh=False)
    assert m.info.has_a1_kernel & 16 and not (m.info.has_a1_kernel & 32) and m.jit_failed() == "", m.describe()
    assert [f.name.endswith(".hsaco") for f in cache.iterdir()] == [True]
