"""Host buffers in flight: nam_hip_batch_submit_f32 / nam_hip_batch_wait_f32 (include/nam_hip.h) — DSP::process (NAM/dsp.h:97)
cut in two for callers that hand buffer k in while buffers k-1 .. k-3 are still being rendered. Every form the library
serves a ticket with — commands of the persistent session with per-command completion (nam_a1_q_kernel, nam_kq_kernel),
commands that complete when the launch has drained its ring (nam_wn_reg_kernel, the LSTM row kernel), copies and launches
on the batch's stream (outside persistent mode), a blocking render kept for the wait (ragged lengths) — against the
blocking entry point on a second batch of the same model fed the same audio, every stream and frame, and stream 0 against
the CPU oracle."""
import numpy as np
import pytest

from conftest import model_path
from signals import stream_bank

pytestmark = pytest.mark.gpu

# (model fixture, streams, persistent, frames per buffer, buffers, kernel expected in the session or None)
CASES = {
    "a1_standard_session_64": ("wavenet_a1_standard", 96, True, 64, 48, "nam_a1_q_kernel"),
    "a1_standard_session_256": ("wavenet_a1_standard", 256, True, 256, 12, "nam_a1_q_kernel"),
    "a1_standard_session_turns_64": ("wavenet_a1_standard", 300, True, 64, 24, "nam_a1_q_kernel"),  # more workgroups than CUs
    "a2_full_session_64": ("A2", 128, True, 64, 40, "nam_kq_kernel"),
    "a2_max_session_128": ("wavenet_a2_max", 64, True, 128, 16, "nam_wn_reg_kernel"),
    "lstm_session_64": ("lstm", 33, True, 64, 24, None),
    "a1_standard_launches_64": ("wavenet_a1_standard", 40, False, 64, 24, None),
    "wavenet_launches_96": ("wavenet", 7, False, 96, 10, None),
    "lstm_launches_64": ("lstm", 9, False, 64, 10, None),
    "a1_standard_session_ragged_100": ("wavenet_a1_standard", 16, True, 100, 9, None),
}


def _feed(batch, x, frames, depth):
    """buffer k in, buffer k - depth out"""
    nb = x.shape[-1] // frames
    ys, tickets = [], []
    for k in range(nb):
        if len(tickets) == depth:
            ys.append(batch.wait(tickets.pop(0)))
        tickets.append(batch.submit(x[:, k * frames:(k + 1) * frames]))
    while tickets:
        ys.append(batch.wait(tickets.pop(0)))
    return np.concatenate(ys, axis=2)


@pytest.mark.parametrize("depth", [1, 4, 16])
@pytest.mark.parametrize("case", sorted(CASES))
def test_tickets_against_the_blocking_call(nam_lib, oracle, case, depth):
    nam = nam_lib
    name, n_streams, persistent, frames, nb, kname = CASES[case]
    if depth != 4 and "session_64" not in case:
        pytest.skip("depths 1 and 16 are covered on the session cases")
    x = stream_bank(n_streams, nb * frames, seed=700 + len(case))
    model = nam.get_dsp(model_path(name), fast_tanh=True)
    ref_b = model.batch(n_streams, frames)
    ref_b.set_persistent(persistent)
    ref_b.Reset(prewarm=True)
    want = ref_b.process_stream(x, frames)
    ref_b.close()
    b = model.batch(n_streams, frames)
    assert bool(b.set_persistent(persistent)) == persistent
    if kname:
        assert b.kernel_name() == kname
    b.Reset(prewarm=True)
    got = _feed(b, x, frames, depth)
    b.close()
    assert got.shape == want.shape and np.isfinite(got).all()
    # (a blocking call of up to four buffers runs nam_a1_p4_kernel where the session runs nam_a1_q_kernel: sums associated differently)
    scale = max(1.0, float(np.abs(want).max()))
    assert float(np.abs(got - want).max()) <= 2e-5 * scale
    # tickets against the CPU oracle itself (not only HIP against HIP): the first stream, the last, and three seeded picks
    rng = np.random.default_rng(len(case) * 31 + depth)
    picks = sorted({0, n_streams - 1} | {int(v) for v in rng.integers(0, n_streams, size=3)})
    for s in picks:
        ref = oracle.get_dsp(model_path(name), fast_tanh=True)
        ref.Reset(48000.0, frames)
        r = ref.process_stream(x[s], frames)[0]
        assert float(np.max(np.abs(r - got[s, 0]))) <= 5e-5 * max(1.0, float(np.max(np.abs(r)))), (case, depth, s)


def test_ticket_rules(nam_lib):
    """NAM_HIP_PIPE_SLOTS in flight, the next is refused until the oldest has been waited for; a ticket is waited for once; waits
    in any order; Reset completes what is in flight and keeps the outputs; blocking calls may be mixed in"""
    nam = nam_lib
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    D = nam.Batch.PIPE_SLOTS
    n, frames = 24, 64
    nb = D + 12
    x = stream_bank(n, nb * frames, seed=77)
    buf = lambda k: x[:, k * frames:(k + 1) * frames]
    ref_b = model.batch(n, frames)
    ref_b.set_persistent(True)
    ref_b.Reset(prewarm=True)
    want = ref_b.process_stream(x, frames)
    ref_b.close()
    scale = max(1.0, float(np.abs(want).max()))
    b = model.batch(n, frames)
    b.set_persistent(True)
    b.Reset(prewarm=True)
    t = [b.submit(buf(k)) for k in range(D)]
    assert t == list(range(D))
    with pytest.raises(nam.NamHipError, match="in flight"):
        b.submit(buf(D))
    y = {}
    y[2] = b.wait(t[2])  # any order
    y[0] = b.wait(t[0])
    with pytest.raises(Exception, match="not in flight"):
        b.wait(t[0])
    t.append(b.submit(buf(D)))  # slot 0 is free again
    assert t[D] == D
    for k in [1] + list(range(3, D + 1)):
        y[k] = b.wait(t[k])
    # a blocking call between tickets (the streams' state carries on)
    y[D + 1] = b.process(buf(D + 1))
    t2 = b.submit(buf(D + 2))
    y[D + 3] = b.process(buf(D + 3))  # with a ticket in flight: it completes first
    y[D + 2] = b.wait(t2)
    got = np.concatenate([y[k] for k in range(D + 4)], axis=2)
    assert float(np.abs(got - want[:, :, :(D + 4) * frames]).max()) <= 2e-5 * scale
    # Reset with tickets in flight: they complete first, their outputs stay
    t4, t5 = b.submit(buf(D + 4)), b.submit(buf(D + 5))
    b.Reset(prewarm=True)
    y45 = np.concatenate([b.wait(t4), b.wait(t5)], axis=2)
    assert float(np.abs(y45 - want[:, :, (D + 4) * frames:(D + 6) * frames]).max()) <= 2e-5 * scale
    # ... and the batch starts over: the first buffers again
    z = _feed(b, x[:, :6 * frames], frames, 4)
    assert float(np.abs(z - want[:, :, :6 * frames]).max()) <= 2e-5 * scale
    with pytest.raises(nam.NamHipError):
        b.submit(np.zeros((n, 1, frames + 1), dtype=np.float32))  # more than max_frames
    b.close()


@pytest.mark.parametrize("case", ["a1_standard", "A2", "lstm", "a1_standard_no_session"])
def test_tickets_mixed_with_everything_else(nam_lib, case):
    """a seeded random walk over the entry points of one batch — submit (whole and ragged lengths), wait (any ticket in flight),
    the blocking call, flush, synchronize, and a Reset now and then — against a second batch that renders the same audio with
    blocking calls only: the buffers must come out the same whatever travelled as a ticket"""
    nam = nam_lib
    name = {"a1_standard": "wavenet_a1_standard", "A2": "A2", "lstm": "lstm", "a1_standard_no_session": "wavenet_a1_standard"}[case]
    persistent = case != "a1_standard_no_session"
    rng = np.random.default_rng(991 + len(case))
    n, max_frames = 40, 256
    model = nam.get_dsp(model_path(name), fast_tanh=True)
    ref_b = model.batch(n, max_frames)
    ref_b.set_persistent(persistent)
    ref_b.Reset(prewarm=True)
    b = model.batch(n, max_frames)
    b.set_persistent(persistent)
    b.Reset(prewarm=True)
    in_flight = {}  # ticket -> the reference rendering of its buffer
    checked = submitted = 0

    def check(got, want):
        nonlocal checked
        scale = max(1.0, float(np.abs(want).max()))
        assert got.shape == want.shape and float(np.abs(got - want).max()) <= 2e-5 * scale
        checked += 1

    for step in range(400):
        op = rng.choice(["submit", "submit", "submit", "wait", "wait", "process", "flush", "sync", "reset"], p=[0.22, 0.22, 0.1, 0.2, 0.1, 0.08, 0.03, 0.03, 0.02])
        if op == "submit" and (submitted - nam.Batch.PIPE_SLOTS) not in in_flight:  # (ticket t + 16 takes the slot of ticket t)
            submitted += 1
            frames = int(rng.choice([64, 64, 128, 192, 256, 100, 37]))
            x = stream_bank(n, frames, seed=int(rng.integers(1 << 30)))
            want = ref_b.process(x)
            in_flight[b.submit(x)] = want
        elif op == "wait" and in_flight:
            t = int(rng.choice(sorted(in_flight)))
            check(b.wait(t), in_flight.pop(t))
        elif op == "process":
            frames = int(rng.choice([64, 128, 256, 90]))
            x = stream_bank(n, frames, seed=int(rng.integers(1 << 30)))
            check(b.process(x), ref_b.process(x))
        elif op == "flush":
            b.flush()
        elif op == "sync":
            b.synchronize()
        elif op == "reset":
            b.Reset(prewarm=True)  # (tickets in flight complete first and keep their outputs)
            ref_b.Reset(prewarm=True)
    for t in sorted(in_flight):
        check(b.wait(t), in_flight[t])
    assert checked > 100
    b.close()
    ref_b.close()


def test_tickets_in_double_precision_buffers(nam_lib):
    """NAM_SAMPLE = double callers (NAM/dsp.h:18-22): nam_hip_batch_submit_f64 / nam_hip_batch_wait_f64 against the blocking
    _f64 call — the same casts in (model.cpp:817) and out (:896), so the same doubles"""
    nam = nam_lib
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    n, frames, nb = 20, 128, 10
    x = stream_bank(n, nb * frames, seed=5).astype(np.float64)
    ref_b = model.batch(n, frames)
    ref_b.set_persistent(True)
    ref_b.Reset(prewarm=True)
    want = ref_b.process_stream(x, frames)
    ref_b.close()
    assert want.dtype == np.float64
    b = model.batch(n, frames)
    b.set_persistent(True)
    b.Reset(prewarm=True)
    got = _feed(b, x, frames, 4)
    b.close()
    assert got.dtype == np.float64 and got.shape == want.shape
    assert float(np.abs(got - want).max()) <= 2e-5 * max(1.0, float(np.abs(want).max()))
    assert np.array_equal(got, got.astype(np.float32).astype(np.float64))  # (float32 values widened, as the reference's cast out)


@pytest.mark.parametrize("linger_us", [None, "0"])
def test_two_ticket_sessions_share_one_device(nam_lib, monkeypatch, linger_us):
    """two batches with tickets in flight at the same time, more workgroups between them than the chip holds at once: each
    session's launch has to make room for the other's (by default after its linger; NAM_HIP_TICKET_LINGER_US=0: at once) —
    every buffer of both against the blocking call"""
    nam = nam_lib
    if linger_us is not None:
        monkeypatch.setenv("NAM_HIP_TICKET_LINGER_US", linger_us)
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    n, frames, nb = 200, 64, 16
    xs = [stream_bank(n, nb * frames, seed=31 + i) for i in range(2)]
    wants = []
    for x in xs:
        ref_b = model.batch(n, frames)
        ref_b.set_persistent(True)
        ref_b.Reset(prewarm=True)
        wants.append(ref_b.process_stream(x, frames))
        ref_b.close()
    bs = [model.batch(n, frames) for _ in range(2)]
    for b in bs:
        b.set_persistent(True)
        b.Reset(prewarm=True)
    tickets, ys = [[], []], [[], []]
    for k in range(nb):
        for i, b in enumerate(bs):
            if len(tickets[i]) == 4:
                ys[i].append(b.wait(tickets[i].pop(0)))
            tickets[i].append(b.submit(xs[i][:, k * frames:(k + 1) * frames]))
    for i, b in enumerate(bs):
        while tickets[i]:
            ys[i].append(b.wait(tickets[i].pop(0)))
        b.close()
        got = np.concatenate(ys[i], axis=2)
        assert float(np.abs(got - wants[i]).max()) <= 2e-5 * max(1.0, float(np.abs(wants[i]).max()))


def test_tickets_survive_control_calls_outside_persistent_mode(nam_lib):
    """Outside persistent mode a ticket is staging + copies + a launch on the batch's stream with an event behind them
    (include/nam_hip.h: control calls complete the tickets in flight first). A ticket submitted BEFORE set_kernel /
    SetSlimmableSize and waited for AFTER it must hold what the blocking call renders for the same buffer."""
    nam = nam_lib
    frames = 64
    # set_kernel between submit and wait (the A1 kernels and the op program share one state layout on this model)
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    x = stream_bank(6, 4 * frames, seed=31)
    ref_b = model.batch(6, frames)
    ref_b.Reset(prewarm=True)
    want = ref_b.process_stream(x, frames)
    ref_b.close()
    b = model.batch(6, frames)
    b.Reset(prewarm=True)
    t0 = b.submit(x[:, :frames])
    t1 = b.submit(x[:, frames:2 * frames])
    b.set_kernel(nam.KERNEL_A1)  # quiesces: both launches have run, their results are kept for the waits
    y1 = b.wait(t1)
    y0 = b.wait(t0)
    y2 = b.process(x[:, 2 * frames:3 * frames])  # the VALU kernel carries on from the same state
    b.set_kernel(nam.KERNEL_AUTO)
    t3 = b.submit(x[:, 3 * frames:])
    y3 = b.wait(t3)
    b.close()
    got = np.concatenate([y0, y1, y2, y3], axis=2)
    assert float(np.abs(got - want).max()) <= 2e-5 * max(1.0, float(np.abs(want).max()))
    # SetSlimmableSize between submit and wait: the ticket holds the OLD width's rendering, the next buffer the new width's
    # from the reset state of that width (slimmable.cpp:489-498)
    model = nam.get_dsp(model_path("slimmable_wavenet"), fast_tanh=True)
    x = stream_bank(5, 2 * frames, seed=32)
    b = model.batch(5, frames)
    b.Reset(prewarm=True)
    ref_b = model.batch(5, frames)
    ref_b.Reset(prewarm=True)
    want0 = ref_b.process(x[:, :frames])
    ref_b.SetSlimmableSize(0.0)
    want1 = ref_b.process(x[:, frames:])
    ref_b.close()
    t0 = b.submit(x[:, :frames])
    b.SetSlimmableSize(0.0)
    y0 = b.wait(t0)
    y1 = b.process(x[:, frames:])
    b.close()
    assert float(np.abs(y0 - want0).max()) <= 1e-6 and float(np.abs(y1 - want1).max()) <= 1e-6


def test_blocking_calls_back_to_back_ride_one_lingering_launch(nam_lib, oracle):
    """nam::DSP::process in a loop (NAM/dsp.h:97; tools/benchmodel.cpp:129-132): from the second call on, the session's launch
    publishes every command and lingers for the next call (round 6). Same audio as a device-resident render, every stream against
    the oracle on three; a foreign device-wide synchronize in the middle returns (bounded by the linger, never a deadlock) and the
    loop goes on; a pause longer than the linger, then more calls; 64-frame and 256-frame calls, one stream and 256."""
    import time
    torch = pytest.importorskip("torch")
    nam = nam_lib
    model = nam.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
    for n_streams, frames, nb in ((256, 64, 40), (1, 64, 40), (32, 256, 12)):
        x = stream_bank(n_streams, nb * frames, seed=900 + n_streams)
        b = model.batch(n_streams, frames)
        assert b.set_persistent(True)
        b.Reset(prewarm=True)
        ys = []
        t_sync = None
        for k in range(nb):
            ys.append(b.process(x[:, k * frames:(k + 1) * frames]))
            if k == nb // 2:
                t0 = time.perf_counter()
                torch.cuda.synchronize()  # (a launch that lingers for the next call holds the device for at most the linger)
                t_sync = time.perf_counter() - t0
            if k == nb // 2 + 4:
                time.sleep(0.002)  # (the launch gives up and leaves; the next call starts one again)
        b.close()
        assert t_sync is not None and t_sync < 0.05, t_sync
        y = np.concatenate(ys, axis=2)
        assert np.isfinite(y).all()
        for s in sorted({0, n_streams - 1, n_streams // 3}):
            ref = oracle.get_dsp(model_path("wavenet_a1_standard"), fast_tanh=True)
            ref.Reset(48000.0, frames)
            r = ref.process_stream(x[s], frames)[0]
            assert float(np.max(np.abs(r - y[s, 0]))) <= 5e-5, (n_streams, frames, s)
