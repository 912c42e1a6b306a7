"""Long sessions of every pipelined kernel inside the driver's GPU suite (bounded: about a minute in all). The reference pins
replay / restart behaviour of its fused path (tools/test/test_a2_fast.cpp:272-363: the same audio through two
implementations, block after block, and again after a Reset); here: a persistent session of >= 200 buffers of real signal —
a flush in the middle, a pause long enough for the resident launch to leave and be started again, more workgroups than the
chip holds at once ("turns") — against ONE ordinary launch of the un-pipelined kernel of the family on the same audio, every
stream and every frame, and six streams (first, last, middle, seeded picks) against the CPU oracle. (Round 3's split-wave bug of nam_kq_kernel only showed in
exactly this kind of run: tools/persist_soak.py, now a test.)"""
import os
import time

import numpy as np
import pytest

from conftest import model_path
from signals import stream_bank

pytestmark = pytest.mark.gpu

# (model fixture, streams, buffers, environment of the session, kernel the session must run)
CASES = {
    "a1_standard_q": ("wavenet_a1_standard", 256, 240, {}, "nam_a1_q_kernel"),
    "a1_standard_q_turns": ("wavenet_a1_standard", 500, 200, {}, "nam_a1_q_kernel"),
    # libm tanh: the ACT_TANH instantiations (session and plain launch) of the headline kernel, tools/render.cpp's setting
    "a1_standard_q_libm_tanh": ("wavenet_a1_standard", 256, 200, {}, "nam_a1_q_kernel", False),
    "a1_lite_q_padded": ("synth_a1_lite", 256, 200, {}, "nam_a1_q_kernel"),  # 12 / 6 zero-padded to the kernel's 16 / 8
    "a1_feather_relu_p4_turns": ("synth_a1_feather_relu", 500, 200, {}, "nam_a1_p4_kernel"),
    "a2_full_kq": ("A2", 256, 240, {}, "nam_kq_kernel"),
    "a2_full_kq_turns": ("A2", 500, 200, {}, "nam_kq_kernel"),
    "a2_max_wn_reg_2_stages": ("wavenet_a2_max", 512, 200, {"NAM_HIP_MAX_STAGES": "2"}, "nam_wn_reg_kernel"),
    "a2_max_wn_reg_4_stages": ("wavenet_a2_max", 512, 200, {"NAM_HIP_MAX_STAGES": "4"}, "nam_wn_reg_kernel"),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_long_session_against_one_launch(nam_lib, oracle, monkeypatch, case):
    torch = pytest.importorskip("torch")
    nam = nam_lib
    name, n_streams, nb, env, kname = CASES[case][:5]
    fast_tanh = CASES[case][5] if len(CASES[case]) > 5 else True
    block = 64
    x = stream_bank(n_streams, nb * block, seed=4100 + len(case))
    model = nam.get_dsp(model_path(name), fast_tanh=fast_tanh)
    xd = torch.from_numpy(x[:, None, :]).cuda()
    # the reference rendering: one launch of the un-pipelined kernel (a fresh batch: its own state)
    monkeypatch.setenv("NAM_HIP_MAX_STAGES", "1")  # no pipelines: the un-pipelined kernel of the family
    ref_b = model.batch(n_streams, nb * block)
    ref_b.Reset(prewarm=True)
    yr = torch.zeros_like(xd)
    ref_b.process_device(xd.data_ptr(), yr.data_ptr(), nb * block, nb * block)
    ref_b.synchronize()
    ref_name = ref_b.kernel_name(nb * block)
    ref_b.close()
    monkeypatch.setenv("NAM_HIP_MAX_STAGES", "0")  # (no cap)
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    b = model.batch(n_streams, block)
    assert b.set_persistent(True) and b.kernel_name() == kname and (ref_name != kname or kname == "nam_wn_reg_kernel")
    b.Reset(prewarm=True)
    yd = torch.zeros_like(xd)
    torch.cuda.synchronize()
    for k in range(nb):
        b.process_device(xd.data_ptr() + k * block * 4, yd.data_ptr() + k * block * 4, block, nb * block)
        if k == nb // 3:
            b.flush()  # the host waits for the commands so far; the session goes on
        if k == (2 * nb) // 3:
            time.sleep(0.003)  # the ring runs empty: the resident launch leaves; the next command starts it again
    b.flush()
    torch.cuda.synchronize()
    b.close()
    d = (yd - yr).abs()
    scale = max(1.0, float(yr.abs().max()))
    bad = (d > 1e-5 * scale).any(dim=2).any(dim=1)
    if bool(bad.any()):
        ids = torch.nonzero(bad)[:, 0].cpu().numpy()
        first = [int(torch.nonzero(d[i, 0] > 1e-5 * scale)[0, 0]) for i in ids[:8]]
        raise AssertionError(f"{case}: {len(ids)} streams differ from the one-launch rendering; first bad frames {first} "
                             f"(buffers {[f // block for f in first]}) of streams {ids[:8].tolist()}; max {float(d.max()):.3e}")
    assert bool(torch.isfinite(yd).all())
    # against the CPU oracle: the first and last stream, the middle ones and a few picked by the case's seed — a stream map wrong
    # in BOTH kernels (they share the host side) would pass the comparison above
    rng = np.random.default_rng(len(case))
    picks = sorted({0, n_streams // 2, n_streams - 1, *rng.integers(0, n_streams, size=3).tolist()})
    tol = 5e-5 if fast_tanh else 1e-4
    for s in picks:
        ref = oracle.get_dsp(model_path(name), fast_tanh=fast_tanh)
        ref.Reset(48000.0, block)
        r = ref.process_stream(x[s], block)[0]
        ys = yd[s, 0].cpu().numpy()
        assert float(np.max(np.abs(r - ys))) <= tol * max(1.0, float(np.max(np.abs(r)))), (case, s)
