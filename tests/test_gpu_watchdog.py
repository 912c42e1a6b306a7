"""The session watchdog (NAM_HIP_PERSIST_TIMEOUT_MS, csrc/api_session.cpp: PersistWatch): a resident launch that makes no
progress for that long is reported as a device failure instead of spinning for ever — and a launch that IS making progress,
however long the host waits for it, is not. The stall is real: another process (tests/helpers/gpu_hog.hip) holds every CU's
LDS, so the session's workgroups (144 KB each) cannot be placed until it ends."""
import os
import shutil
import subprocess
import time

import numpy as np
import pytest

from conftest import ROOT, model_path
from signals import stream_bank

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hog(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path_factory.mktemp("hog") / "gpu_hog")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-o", exe, os.path.join(ROOT, "tests", "helpers", "gpu_hog.hip")],
                          stderr=subprocess.DEVNULL)
    return exe


@pytest.mark.timeout(300)
def test_watchdog_reports_a_launch_that_cannot_run(nam_lib, oracle, hog, monkeypatch):
    nam = nam_lib
    n, frames = 32, 64
    x = stream_bank(n, 8 * frames, seed=41)
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    monkeypatch.setenv("NAM_HIP_PERSIST_TIMEOUT_MS", "60")
    b = model.batch(n, frames)
    assert b.set_persistent(True)
    b.Reset(prewarm=True)
    y0 = b.process(x[:, :frames])  # (the session works)
    b.flush()
    p = subprocess.Popen([hog, "1500"], stdout=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "running"
    t0 = time.perf_counter()
    with pytest.raises(nam.NamHipError, match="no progress for 60 ms"):
        b.process(x[:, frames:2 * frames])
    waited = time.perf_counter() - t0
    assert 0.05 <= waited < 1.0, waited  # (it gave up after the timeout, not when the other process ended)
    assert p.wait(timeout=60) == 0
    b.close()  # (the launch has run by now or runs now: closing the batch must not hang)
    monkeypatch.delenv("NAM_HIP_PERSIST_TIMEOUT_MS")
    # the library is intact: a new batch renders the same audio, against the oracle
    b2 = model.batch(n, frames)
    b2.set_persistent(True)
    b2.Reset(prewarm=True)
    y = b2.process_stream(x, frames)
    b2.close()
    assert np.array_equal(y[:, :, :frames], y0)
    for s in (0, n - 1):
        ref = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
        ref.Reset(48000.0, frames)
        assert float(np.max(np.abs(ref.process_stream(x[s], frames)[0] - y[s, 0]))) <= 5e-5


@pytest.mark.timeout(300)
def test_watchdog_leaves_a_busy_launch_alone(nam_lib, monkeypatch):
    """A wait far longer than the timeout, with the launch consuming commands all the time: not a failure (the clock restarts
    whenever a workgroup's progress word moves)."""
    torch = pytest.importorskip("torch")
    nam = nam_lib
    n, frames, calls = 256, 2048, 28  # 28 x 32 = 896 commands in the ring: ~5 ms of work behind ONE flush
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    monkeypatch.setenv("NAM_HIP_PERSIST_TIMEOUT_MS", "1")
    b = model.batch(n, frames)
    assert b.set_persistent(True)
    b.Reset(prewarm=True)
    xd = torch.from_numpy(stream_bank(n, frames, seed=42)[:, None, :]).cuda()
    yd = torch.zeros_like(xd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        b.process_device(xd.data_ptr(), yd.data_ptr(), frames, frames)
    b.flush()
    waited = time.perf_counter() - t0
    b.close()
    assert waited > 0.003 and bool(torch.isfinite(yd).all())
