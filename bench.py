#!/usr/bin/env python3
"""
bench.py — real-time audio streams (xRT) at 48 kHz on MI355X: BASELINE.json's metric on its configs.

    python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}]

  --config 2 (default)  wavenet_a1_standard.nam, 256 concurrent streams per GPU, buffer = 64   (BASELINE configs[1], headline)
  --config 3            lstm.nam, 1,024 concurrent streams per GPU                             (configs[2])
  --config 4            wavenet_a2_max.nam (+ FiLM, nested condition_dsp), 512 streams per GPU (configs[3]: 4,096 over 8 GPUs)
  --config 5            slimmable_wavenet.nam, 768 streams per GPU at mixed widths             (configs[4])

A "step" is one pass of the hot path over one batch: every stream of the batch advances by one 64-frame buffer
(`DSP::process(in, out, 64)` for all streams at once). Inputs are resident in HBM before the timed region. After W
untimed warm-up steps a timed region is EXACTLY K steps between barrier + torch.cuda.synchronize() pairs, MAX over
ranks; the region is repeated --reps times (default 11, protocol of the reference's benchmark_wavenet_a1.sh:10) and
`value` is the MEDIAN region (every region's time is in the line). Weak scaling: every GPU owns `--streams`
independent streams, sharded per width class (no data-path collective; RCCL only broadcasts the model text once,
scatters the input bank before and gathers the rendered tail after the timed regions).

Launch modes (the kernel is the same):
  --launch block     one kernel launch per 64-frame step, K launches enqueued back to back
                     (the real-time serving shape: buffer = 64 samples)            [default]
  --launch resident  ONE launch walks all K steps of the resident signal (offline re-amp shape)

Prints ONE JSON line (rank 0): contract keys + `roofline` (algorithmic bytes of SURVEY.md §8d against the 8 TB/s HBM
peak, with the fp32 fraction, the PMC-measured HBM bytes and the LDS counters of the dominant kernel from
profiles/traffic.json) + `cpu_baseline` (the CPU oracle on the box's host cores) + `latency_us` (per-launch
min / p50 / p99 / p99.9, as tools/bench_a2_fast.cpp:274-296 prints them) + `fast_tanh_off` and `zeros_input` side runs.

--dry-run executes the same scatter -> steps -> gather -> all_reduce(MAX) code on CPU tensors over gloo with a stub
in place of the kernels (tests/test_bench_dry_run.py): the distributed branch is exercised without GPUs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SR = 48000.0
FP32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 vector == FP32 (f32-in) MFMA
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
HBM_ACHIEVABLE_GBS = 6300.0
LDS_PEAK_TBS = 150.0  # MI355X_MICROARCH.md, LDS: ~150 TB/s for ds_read_b64/b128 with every CU streaming at ~2.4 GHz
LDS_CLOCK_HZ = 2.4e9
# issue-port cycles per non-matrix vector instruction of a SIMD when an entry of profiles/traffic.json does not carry its own
# figure (`issue_cycles_per_valu_inst`, measured at the kernel's own occupancy and mix by tools/src/valu_rate.hip): a wave64
# v_fma_f32 is 2.28 cycles at four waves per SIMD, 3.7 - 4.1 next to fp32 matrix instructions, 8.6 for a lone wave
# (profiles/r05/valu_rate_microbench.txt; round 4's three-wave rows timed wave 0 only and read low)
ISSUE_CPI_DEFAULT = 4.0


def wavenet_history_bytes_per_sample(cfg: dict) -> int:
    """Algorithmic history traffic per stream-sample, SURVEY.md §8d-ii's figure: per dilated layer
    (K tap reads + 1 write) x C channels x 4 B. wavenet_a1_standard -> 3,840. (The kernels never re-read
    the current tap, so what must actually move is (K-1) reads + 1 write = 2,880 B; the contract
    figure is kept so numbers stay comparable with SURVEY.md / BASELINE.md.)"""
    total = 0
    if cfg.get("condition_dsp"):
        total += wavenet_history_bytes_per_sample(cfg["condition_dsp"]["config"])
    for lc in cfg["layers"]:
        n = len(lc["dilations"])
        ks = lc.get("kernel_sizes") or [lc["kernel_size"]] * n
        total += sum((k + 1) * lc["channels"] * 4 for k in ks if k > 1)
    return total


def measured_traffic(kernel: str, model: str, streams: int, block: int, launch: str):
    """Counters of the dominant kernel from committed rocprofv3 PMC passes (profiles/traffic.json), if the profiled
    configuration matches this run: keyed by the kernel FUNCTION name the library reports for the batch
    (nam_hip_batch_kernel_name), the model and the launch shape — a K-tap run can never pick up an a1 entry."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            for e in json.load(f)["entries"]:
                if (e["kernel"], e.get("model", "wavenet_a1_standard"), e["streams"], e["block"], e["launch"]) == (
                        kernel, model, streams, block, launch):
                    return e
    except Exception:
        pass
    return None


def wavenet_macs_per_sample(cfg: dict) -> int:
    """Algorithmic MACs per stream-sample of a WaveNet config (dense, grouped convs counted per group).
    wavenet_a1_standard -> 13,320."""
    macs = 0
    if cfg.get("condition_dsp"):
        macs += wavenet_macs_per_sample(cfg["condition_dsp"]["config"])
    for lc in cfg["layers"]:
        C = lc["channels"]
        B = lc.get("bottleneck", C)
        cs = lc["condition_size"]
        n = len(lc["dilations"])
        ks = lc.get("kernel_sizes") or [lc["kernel_size"]] * n
        gm = lc.get("gating_mode")
        if gm is None:
            gm = ["gated" if lc.get("gated") else "none"] * n
        elif isinstance(gm, str):
            gm = [gm] * n
        macs += lc["input_size"] * C
        l1 = lc.get("layer1x1", {"active": True, "groups": 1})
        h1 = lc.get("head1x1", {"active": False})
        for l in range(n):
            zc = 2 * B if gm[l] != "none" else B
            macs += ks[l] * C * zc // lc.get("groups_input", 1)
            macs += cs * zc // lc.get("groups_input_mixin", 1)
            if l1.get("active", True):
                macs += B * C // l1.get("groups", 1)
            if h1.get("active"):
                macs += B * h1["out_channels"] // h1.get("groups", 1)
            dims = {"conv_pre_film": C, "conv_post_film": zc, "input_mixin_pre_film": cs, "input_mixin_post_film": zc,
                    "activation_pre_film": zc, "activation_post_film": B, "layer1x1_post_film": C,
                    "head1x1_post_film": h1.get("out_channels", 0)}
            for key, d in dims.items():
                f = lc.get(key)
                if f and f.get("active", True):
                    macs += cs * (2 if f.get("shift", True) else 1) * d // f.get("groups", 1)
        head_in = h1["out_channels"] if h1.get("active") else B
        if lc.get("head"):
            macs += head_in * lc["head"]["out_channels"] * lc["head"]["kernel_size"]
        else:
            macs += head_in * lc["head_size"]
    return macs


def active_model_json(path: str) -> dict:
    """The .nam document that actually runs: a SlimmableContainer starts on its last submodel (container.cpp:49)."""
    with open(path) as f:
        j = json.load(f)
    while j["architecture"] == "SlimmableContainer":
        j = j["config"]["submodels"][-1]["model"]
    return j


def model_macs(path: str) -> int:
    j = active_model_json(path)
    if j["architecture"] == "WaveNet":
        return wavenet_macs_per_sample(j["config"])
    c = j["config"]
    H, I, L = c["hidden_size"], c["input_size"], c["num_layers"]
    return sum(4 * H * ((I if l == 0 else H) + H) for l in range(L)) + H * c.get("out_channels", 1)


def cpu_baseline(model_path: str, fast_tanh: bool, block: int, target_seconds: float = 12.0, extras: bool = True):
    """CPU oracle ("port": our Eigen-free restatement of the reference path, bit-exact with the reference's own sources
    built on a scalar Eigen stand-in — oracle/_ref, timed alongside as a floor: real Eigen is not in this image) on ONE
    host core, benchmodel protocol (64-frame blocks, Reset + prewarm first), on a bounded sample of the same workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nam_oracle
    from signals import two_tone
    fast_so = os.path.join("/tmp", f"libnam_oracle_fast_{os.getpid()}.so")
    kind_flags = "-Ofast -march=native"
    try:
        nam_oracle.build_fast(fast_so)
        nam_oracle.use_library(fast_so)
    except Exception:
        kind_flags = "-O3 -march=x86-64-v3 -ffp-contract=off"
    m = nam_oracle.get_dsp(model_path, fast_tanh=fast_tanh)
    m.Reset(SR, block)
    probe = two_tone(int(SR))  # 1 s
    t0 = time.perf_counter()
    m.process_stream(probe, block)
    dt = time.perf_counter() - t0
    secs_audio = max(2.0, min(120.0, target_seconds / max(dt, 1e-6)))
    x = two_tone(int(secs_audio * SR))
    t0 = time.perf_counter()
    m.process_stream(x, block)
    dt = time.perf_counter() - t0
    try:
        os.remove(fast_so)
    except OSError:
        pass
    out = {
        "value": round(len(x) / SR / dt, 3), "unit": "xRT (48 kHz real-time streams)", "cores": 1, "kind": "port",
        "sample": f"1 stream x {secs_audio:.1f} s of two-tone audio in {block}-frame blocks after Reset+prewarm, "
                  f"oracle/nam_oracle.c built {kind_flags}, {dt:.2f} s of CPU",
    }
    # BASELINE.md section 3's PRIMARY figure is the -O2 build ("g++ -O2 -march=native, and separately -Ofast")
    try:
        o2_so = os.path.join("/tmp", f"libnam_oracle_o2_{os.getpid()}.so")
        nam_oracle.build_fast(o2_so, "-O2")
        nam_oracle.use_library(o2_so)
        m2 = nam_oracle.get_dsp(model_path, fast_tanh=fast_tanh)
        m2.Reset(SR, block)
        x2 = x[:max(int(SR), int(len(x) * min(1.0, (target_seconds / 3.0) / max(dt, 1e-6))))]
        t0 = time.perf_counter()
        m2.process_stream(x2, block)
        dt2 = time.perf_counter() - t0
        out["O2"] = {"value": round(len(x2) / SR / dt2, 3), "cores": 1,
                     "note": f"same port built -O2 -march=native (BASELINE.md section 3 primary), {len(x2) / SR:.1f} s of audio, {dt2:.2f} s of CPU"}
        os.remove(o2_so)
    except Exception as e:
        out["O2"] = {"error": str(e)[:200]}
    if not extras:
        return out
    # alongside: every host core at once (SURVEY 8d asks for the multi-core figure next to the single-thread one):
    # one worker process per core, one stream each, same protocol; the aggregate is what the host could sustain
    try:
        import subprocess
        # (capped at 32 workers: the GPU box shows 256 logical CPUs but its container sustains ~13 cores' worth of
        # work — 256 workers took 49 s for an aggregate of 196 xRT)
        n_workers = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 32))
        secs_w = max(2.0, min(10.0, 4.0 / max(dt / max(secs_audio, 1e-6), 1e-6)))  # a few seconds of CPU per worker
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", model_path, str(int(fast_tanh)), str(block), str(secs_w), fast_so]
        nam_oracle.build_fast(fast_so)
        t0 = time.perf_counter()
        procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(n_workers)]
        rates = []
        for pr in procs:
            o, _ = pr.communicate(timeout=180)
            if pr.returncode == 0 and o.strip():
                rates.append(float(o.strip().splitlines()[-1]))
        if rates:
            out["all_host_cores"] = {"value": round(sum(rates), 1), "cores": len(rates),
                                     "note": f"{len(rates)} worker processes x 1 stream x {secs_w:.1f} s each, run concurrently "
                                             f"({time.perf_counter() - t0:.1f} s wall incl. start-up); sum of the workers' own rates"}
        os.remove(fast_so)
    except Exception as e:
        out["all_host_cores"] = {"error": str(e)[:200]}
    # alongside: the reference's own sources (oracle/_ref, prebuilt where /root/reference exists). Their Eigen calls
    # run on a scalar stand-in, so this is a floor for the reference, not its real speed; the faster of the two
    # (the port) stays the reported baseline.
    try:
        import nam_ref
        if os.path.exists(nam_ref.LIB):
            r = nam_ref.get_dsp(model_path, fast_tanh)
            r.Reset(SR, block)
            xr = two_tone(int(min(secs_audio, 20.0) * SR))
            t0 = time.perf_counter()
            r.process_stream(xr, block)
            out["reference_sources_on_eigen_stand_in"] = {"value": round(len(xr) / SR / (time.perf_counter() - t0), 3), "cores": 1,
                                                            "note": "oracle/_ref/libnam_ref.so, -O2, scalar Eigen stand-in"}
    except Exception as e:  # the checker library is optional
        out["reference_sources_on_eigen_stand_in"] = {"error": str(e)[:200]}
    # A2.nam: what the reference itself runs for this shape is its fused wavenet/a2_fast.cpp path (fixed-size matrices: the
    # scalar Eigen stand-in costs it little) — timed too where the checker library travelled
    try:
        import nam_ref
        if os.path.basename(model_path) == "A2.nam" and os.path.exists(getattr(nam_ref, "LIB_A2FAST", "")):
            r = nam_ref.get_dsp(model_path, fast_tanh, a2_fast=True)
            r.Reset(SR, block)
            xr = two_tone(int(min(secs_audio, 20.0) * SR))
            t0 = time.perf_counter()
            r.process_stream(xr, block)
            out["reference_a2_fast_path"] = {"value": round(len(xr) / SR / (time.perf_counter() - t0), 3), "cores": 1,
                                             "note": "oracle/_ref/libnam_ref_a2fast.so = the reference's wavenet/a2_fast.cpp, unmodified, -O2"}
    except Exception as e:
        out["reference_a2_fast_path"] = {"error": str(e)[:200]}
    return out



CONFIGS = {
    2: dict(model="wavenet_a1_standard", streams=256, slim_mix=False, name="BASELINE.json configs[1]"),
    3: dict(model="lstm", streams=1024, slim_mix=False, name="BASELINE.json configs[2]"),
    4: dict(model="wavenet_a2_max", streams=512, slim_mix=False, name="BASELINE.json configs[3] (4,096 streams over 8 GPUs = 512 per GPU)"),
    5: dict(model="slimmable_wavenet", streams=768, slim_mix=True, name="BASELINE.json configs[4] (ratios 0.0 / 0.34 / 0.67 / 1.0 -> widths 1, 2, 3, 3)"),
}
SLIM_RATIOS = (0.0, 0.34, 0.67, 1.0)


def cpu_worker(argv):
    """One worker of cpu_baseline's all-cores figure: prints its own xRT for one stream (no torch import)."""
    import numpy as np  # noqa: F401
    model_path, fast_tanh, block, secs, lib = argv[0], bool(int(argv[1])), int(argv[2]), float(argv[3]), argv[4]
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nam_oracle
    from signals import two_tone
    if os.path.exists(lib):
        nam_oracle.use_library(lib)
    m = nam_oracle.get_dsp(model_path, fast_tanh=fast_tanh)
    m.Reset(SR, block)
    x = two_tone(int(secs * SR))
    t0 = time.perf_counter()
    m.process_stream(x, block)
    print(len(x) / SR / (time.perf_counter() - t0), flush=True)


def run_other_configs(args):
    """Brief runs of BASELINE.json configs 3, 4, 5 (+ A2-Full) for the default line (`other_configs`): one subprocess each (a failure
    of one cannot take the headline with it), `--brief`: 500 timed steps x 11 regions, no side runs, ~2 s CPU baseline."""
    import subprocess
    keep = ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "max_abs_err_vs_oracle", "parity_detail", "cpu_baseline",
            "finite", "region_us")
    res = {}
    # "A2": not a BASELINE.json config — the reference's own flagship shape (A2.nam's A2-Full submodel, what its fused
    # wavenet/a2_fast.cpp path is written for), 256 streams, same protocol
    # "2_steady": the headline configuration itself in this protocol (regions of 500 steps: what a session that lives longer
    # than the driver's 20-step regions sustains — a region pays launch, prologue and the first buffer's way through the
    # pipeline once)
    # "3_long": config 3 with a long block ("block 64, and a long-block variant, e.g. 4,096, since offline re-amping allows
    # it"): ONE launch walks 4,096 frames of every stream (`--launch resident --steps 64`)
    for c in ("2_steady", 3, "3_long", 4, 5, "A2"):
        sel = (["--model", "A2", "--streams", "256"] if c == "A2" else ["--config", "3", "--launch", "resident"] if c == "3_long"
               else ["--config", "2" if c == "2_steady" else str(c)])
        steps, warm = ("64", "64") if c == "3_long" else ("500", "50")
        cmd = [sys.executable, os.path.abspath(__file__)] + sel + ["--gpus", "1", "--steps", steps, "--warmup", warm,
               "--brief", "--full-line", "--persistent", str(args.persistent), "--fast-tanh", str(args.fast_tanh)]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                res[str(c)] = {"error": (p.stderr or p.stdout)[-400:], "rc": p.returncode}
                continue
            j = json.loads(line[-1])
            r = {k: j.get(k) for k in keep}
            r["workload"] = ("A2.nam (A2-Full), 256 streams, buffer 64" if c == "A2" else
                             CONFIGS[2]["name"] + ", regions of 500 steps" if c == "2_steady" else
                             CONFIGS[3]["name"] + ", one launch per 4,096-frame block" if c == "3_long" else CONFIGS[c]["name"])
            r["run_s"] = round(time.perf_counter() - t0, 1)
            res[str(c)] = r
        except Exception as e:  # noqa: BLE001
            res[str(c)] = {"error": str(e)[:400]}
    return res


class HipEngine:
    """The product path: one nam_hip batch on this rank's GPU, launched on a dedicated HIP stream."""

    def __init__(self, nam, torch, model, n_streams, block, local_rank, kernel, classes, persistent=False):
        self.torch, self.nam = torch, nam
        self.dev = torch.device("cuda", local_rank)
        self.batch = model.batch(n_streams, block, device=local_rank)
        if kernel != "auto":
            self.batch.set_kernel({"generic": nam.KERNEL_GENERIC, "a1": nam.KERNEL_A1, "a1_mfma": nam.KERNEL_A1_MFMA, "a1_il": nam.KERNEL_A1_IL,
                                   "wn_reg": nam.KERNEL_WN_REG}[kernel])
        self.batch.Reset(prewarm=True)
        if classes is not None:  # mixed widths: stream i of this rank runs at ratio SLIM_RATIOS[classes[i]]
            for c in sorted(set(classes)):
                self.batch.SetSlimmableSize(SLIM_RATIOS[c], [i for i, ci in enumerate(classes) if ci == c])
        # a dedicated (non-null) HIP stream: the kernels are launched on it through the C ABI and the HIP events that
        # time them are recorded on the same stream
        self.stream = torch.cuda.Stream(self.dev)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        assert self.stream.cuda_stream != 0
        self.block = block
        # persistent block mode: one resident launch consumes one doorbell per step (include/nam_hip.h)
        self.persistent = bool(persistent) and self.batch.set_persistent(True)

    def bind(self, x, y):
        self.xp, self.yp, self.T = x.data_ptr(), y.data_ptr(), x.shape[2]

    def run_steps(self, first, count, launch):
        sh, b = self.stream.cuda_stream, self.block
        if launch == "block":
            for s in range(first, first + count):
                self.batch.process_device(self.xp + s * b * 4, self.yp + s * b * 4, b, self.T, sh)
        else:
            self.batch.process_device(self.xp + first * b * 4, self.yp + first * b * 4, count * b, self.T, sh)

    def event(self):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record(self.stream)
        return e

    def elapsed_ms(self, e0, e1):
        return e0.elapsed_time(e1)

    def sync(self):
        # The contract's device-wide synchronize FIRST, the session's flush behind it. Every command of the region is in the ring
        # by now (host stores, fenced) and the session's launch leaves by itself once it has drained the ring, so the synchronize
        # returns when the region's work is done — and, called while the launch is still running, its marker is already queued
        # behind the launch: it returns ~2 us after the launch retires. Called AFTER a flush that has polled the completion words
        # (round 4's order) the same synchronize pushes its marker through an idle queue: 11 us for nothing
        # (tools/src/sync_tail.hip, profiles/r05/sync_tail.txt: launch -> synchronized 120.5 vs 129.9 us). The flush then finds
        # every workgroup's count published (it would start the launch again, and wait for it, had a command been missed).
        if self.persistent and os.environ.get("NAM_BENCH_FLUSH_FIRST") == "1":  # (round 4's order, kept for same-box A/B runs)
            self.batch.flush(self.stream.cuda_stream)
            self.t_flushed = time.perf_counter()
            self.torch.cuda.synchronize(self.dev)
            return
        self.torch.cuda.synchronize(self.dev)
        if self.persistent:
            self.batch.flush(self.stream.cuda_stream)
            self.t_flushed = time.perf_counter()

    def kernel_name(self):
        return self.batch.kernel_name()

    def close(self):
        self.batch.close()


class StubEngine:
    """--dry-run: stands in for the kernels on CPU tensors (y = 0.5 x on the window) so that the distributed plumbing
    of this file — model-text broadcast, per-class sharding, scatter, the timed regions with their barriers and
    all_reduce(MAX), gather — runs under gloo without a GPU. Never used for a measurement."""

    def __init__(self, torch, block):
        self.torch, self.block = torch, block

    def bind(self, x, y):
        self.x, self.y = x, y

    def run_steps(self, first, count, launch):
        a, b = first * self.block, (first + count) * self.block
        self.y[:, :, a:b] = 0.5 * self.x[:, :1, a:b]

    def event(self):
        return time.perf_counter()

    def elapsed_ms(self, e0, e1):
        return (e1 - e0) * 1e3

    def sync(self):
        pass

    def kernel_name(self):
        return "stub"

    def close(self):
        pass


def host_io(model_path: str, n_streams: int, fast_tanh: bool, budget_s: float = 0.3):
    """Host buffers in, host buffers out (`host_io` of the default line; never `value`: the timed region of `value` starts with
    the inputs in HBM): 256 streams of the headline model through the C ABI's two host-buffer forms — the blocking
    nam_hip_batch_process_f32 (one buffer at a time: copy in, commands, wait, copy out) and the ticketed
    nam_hip_batch_submit_f32 / nam_hip_batch_wait_f32 with NAM_HIP_PIPE_SLOTS buffers in flight (the resident launch renders
    buffer k while the host copies k + 1 in and k - 1 out). Raw ctypes calls on preallocated arrays: the loop costs ~1 us
    of interpreter per call."""
    import ctypes
    import numpy as np
    import neuralampmodelercore_amd as nam
    model = nam.get_dsp(model_path, fast_tanh=fast_tanh)
    L = nam.load_library()
    res = {}
    rng = np.random.default_rng(5)
    for frames in (64, 256, 1024):
        b = model.batch(n_streams, frames)
        b.set_persistent(True)
        b.Reset(prewarm=True)
        D = nam.Batch.PIPE_SLOTS
        xs = [np.ascontiguousarray(rng.uniform(-0.5, 0.5, size=(n_streams, 1, frames)).astype(np.float32)) for _ in range(D)]
        ys = [np.zeros((n_streams, 1, frames), dtype=np.float32) for _ in range(D)]
        xp = [x.ctypes.data_as(ctypes.c_void_p) for x in xs]
        yp = [y.ctypes.data_as(ctypes.c_void_p) for y in ys]
        h = b._h
        entry = {}
        for mode in ("blocking", "tickets", "tickets_depth_1"):
            def run(n):
                if mode == "tickets_depth_1":  # submit, wait at once: what a blocking call routed through the ticket path would cost
                    t1 = ctypes.c_int64(-1)
                    for i in range(n):
                        if L.nam_hip_batch_submit_f32(h, xp[i % D], frames, ctypes.byref(t1)) != 0 or L.nam_hip_batch_wait_f32(h, t1, yp[i % D]) != 0:
                            raise RuntimeError(L.nam_hip_last_error().decode())
                elif mode == "blocking":
                    for i in range(n):
                        if L.nam_hip_batch_process_f32(h, xp[i % D], yp[i % D], frames) != 0:
                            raise RuntimeError(L.nam_hip_last_error().decode())
                else:
                    t = [ctypes.c_int64(-1) for _ in range(D)]
                    for i in range(n):
                        k = i % D
                        if i >= D and L.nam_hip_batch_wait_f32(h, t[k], yp[k]) != 0:
                            raise RuntimeError(L.nam_hip_last_error().decode())
                        if L.nam_hip_batch_submit_f32(h, xp[k], frames, ctypes.byref(t[k])) != 0:
                            raise RuntimeError(L.nam_hip_last_error().decode())
                    for i in range(max(n - D, 0), n):
                        if L.nam_hip_batch_wait_f32(h, t[i % D], yp[i % D]) != 0:
                            raise RuntimeError(L.nam_hip_last_error().decode())
            run(64)
            n = 64
            t0 = time.perf_counter()
            run(n)
            dt = time.perf_counter() - t0
            n = int(min(max(n * budget_s / max(dt, 1e-6), 64), 200000))
            dts = []
            for _ in range(2):  # (the better of two passes: the host's copies share the box with whatever else it runs)
                t0 = time.perf_counter()
                run(n)
                dts.append(time.perf_counter() - t0)
            dt = min(dts)
            entry[mode] = {"value": round(n_streams * frames * n / SR / dt, 1), "us_per_call": round(dt / n * 1e6, 2), "calls": n,
                           "us_per_call_passes": [round(d_ / n * 1e6, 2) for d_ in dts]}
        entry["finite"] = bool(all(np.isfinite(y).all() for y in ys))
        res[f"{frames}_frames"] = entry
        b.close()
    res["unit"] = "xRT, host buffer to host buffer"
    res["in_flight"] = nam.Batch.PIPE_SLOTS
    res["note"] = ("PCIe-inclusive: the host writes each input through the BAR window and reads each output from host-mapped memory "
                   "the resident launch stores to; `tickets` = nam_hip_batch_submit_f32 / nam_hip_batch_wait_f32 with `in_flight` buffers "
                   "between submit and wait, `blocking` = nam_hip_batch_process_f32. Not `value`.")
    return res


def percentile(sorted_vals, q):
    if not sorted_vals:
        return None
    i = min(len(sorted_vals) - 1, max(0, int(round(q * (len(sorted_vals) - 1)))))
    return sorted_vals[i]


LINE_LIMIT = 6000  # bytes: the driver keeps an 8 KB tail of stdout and parses the last line out of it (round 4's 25 KB line was cut)


def _brief_config(r):
    """One other configuration of the default run, in a few numbers."""
    if not isinstance(r, dict) or "value" not in r:
        return {"error": str((r or {}).get("error", "no line"))[:120]}
    rf = r.get("roofline") or {}
    return {"value": r["value"], "ms_per_step": r.get("ms_per_step"), "kernel": (r.get("config") or {}).get("kernel") or r.get("kernel"),
            "bound": rf.get("bound"), "frac": rf.get("frac"), "floor_frac": rf.get("floor_frac"),
            "max_abs_err_vs_oracle": r.get("max_abs_err_vs_oracle"),
            "cpu_baseline": (r["cpu_baseline"].get("value") if isinstance(r.get("cpu_baseline"), dict) else r.get("cpu_baseline"))}


def compact_line(out: dict, full_path) -> dict:
    """The ONE line the driver parses: contract keys, `roofline`, `cpu_baseline` and a handful of figures, numbers and short
    names only. Everything else this file measures (every region's time, prose notes, the other configurations' own
    rooflines, host_io by buffer size) is in the full record: `full_path` (gpurun_out/bench_full.json), or --full-line."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "finite", "max_abs_err_vs_oracle", "parity_per_rank", "parity_streams_checked")
    line = {k: out[k] for k in keep if k in out}
    c = out.get("config") or {}
    line["config"] = {k: c[k] for k in ("workload", "baseline_config", "streams_per_gpu", "block", "launch", "kernel",
                                         "persistent_block_mode", "sharding") if k in c}
    rf = out.get("roofline") or {}
    line["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "floor_frac",
                                                "floor_parts_us", "issue_cycles_per_valu_inst", "counter_hbm_frac", "source")}
    if rf.get("hbm_contract"):
        line["roofline"]["hbm_contract"] = {k: rf["hbm_contract"].get(k) for k in ("achieved", "unit", "frac", "label")}
    if rf.get("lds"):
        line["roofline"]["lds"] = {k: rf["lds"].get(k) for k in ("achieved", "peak", "frac", "array_busy_frac")}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
        line["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:160]
        for k in ("O2", "all_host_cores", "reference_sources_on_eigen_stand_in"):
            if isinstance(cb.get(k), dict) and "value" in cb[k]:
                line["cpu_baseline"][k] = {"value": cb[k]["value"], "cores": cb[k].get("cores")}
    rep = out.get("repetitions") or {}
    line["repetitions"] = {k: rep.get(k) for k in ("n", "untimed_in_front", "ms_per_step_min", "ms_per_step_max")}
    if out.get("region_us"):
        line["region_us"] = out["region_us"]
    if out.get("latency_us"):
        line["latency_us"] = {k: out["latency_us"].get(k) for k in ("kernel", "min", "p50", "p99", "p99_9")}
    if out.get("resident_launch"):
        line["resident_launch"] = {k: out["resident_launch"].get(k) for k in ("value", "ms_per_step")}
    for k in ("zeros_input", "fast_tanh_off", "fast_tanh_on", "wav_input"):
        if isinstance(out.get(k), dict):
            line[k] = {q: out[k].get(q) for q in ("value", "ms_per_step", "kernel", "max_abs_err_vs_oracle", "error") if q in out[k]}
    if isinstance(out.get("steady_state"), dict):
        line["steady_state"] = {k: out["steady_state"].get(k) for k in ("value", "ms_per_step", "steps_per_region", "max_abs_err_vs_oracle", "floor_frac", "compute_frac")}
    if isinstance(out.get("other_configs"), dict):
        line["other_configs"] = {k: _brief_config(v) for k, v in out["other_configs"].items() if k != "2_steady"}
    hio = out.get("host_io")
    if isinstance(hio, dict):
        line["host_io"] = ({"error": str(hio["error"])[:120]} if "error" in hio else
                           {k: {m: v[m]["value"] for m in ("blocking", "tickets") if m in v} for k, v in hio.items() if isinstance(v, dict)})
    if out.get("parity_detail"):
        line["parity_detail"] = out["parity_detail"]
    line["full_record"] = full_path
    # a guard, not a plan: shed the optional blocks (least important first) should the line ever outgrow the limit
    for k in ("host_io", "wav_input", "zeros_input", "resident_launch", "latency_us", "region_us", "repetitions", "other_configs"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(k, None)
    return line


def write_full_record(out: dict):
    """The whole record next to the profiles' scratch (gpurun_out/ travels back from a gpurun box); None if nowhere to write."""
    for d in (os.path.join(ROOT, "gpurun_out"), "/tmp"):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_full.json")
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            return os.path.relpath(path, ROOT) if d.startswith(ROOT) else path
        except OSError:
            continue
    return None


def wav_fidelity_run(fresh_engine, model, n_streams, block, dev, oracle_err, fast_tanh, torch, np, seconds=10.0, passes=10):
    """SURVEY.md 8(d), config 2's input recipe: see the call site. The whole first pass of stream 0 (the WAV, looped: 10 s) and the
    first 2,560 frames of the last stream are checked against the oracle; the passes are timed like the headline's regions
    (fence, K steps, fence) with K = every 64-frame buffer of the 10 s."""
    from signals import read_wav_mono24, two_tone
    wav, sr = read_wav_mono24(os.path.join(ROOT, "tests", "golden", "audio", "input.wav"))
    assert int(sr) == int(SR)
    steps = int(np.ceil(seconds * SR / block))
    T = steps * block
    bank = np.empty((n_streams, T), dtype=np.float32)
    bank[0] = np.tile(wav, T // len(wav) + 1)[:T]
    for s_ in range(1, n_streams):
        bank[s_] = two_tone(T, 1.0 + s_ / n_streams)  # (0.25 + 0.10 = the recipe's 0.35 peak)
    xw = torch.from_numpy(bank[:, None, :]).to(dev)
    yw = torch.zeros_like(xw)
    e = fresh_engine(model)
    e.bind(xw, yw)
    ts = []
    err = None
    for p_ in range(passes + 1):  # (the first pass is untimed: it is the one checked)
        e.sync()
        t0 = time.perf_counter()
        e.run_steps(0, steps, "block")
        e.sync()
        if p_ == 0:
            rows = [0, n_streams - 1]
            got0 = yw[0, 0].cpu().numpy()
            gotl = yw[n_streams - 1, 0, :2560].cpu().numpy()
            err = max(oracle_err(fast_tanh, bank[None, 0:1, :], got0[None], [0]),
                      oracle_err(fast_tanh, bank[None, n_streams - 1:n_streams, :2560], gotl[None], [n_streams - 1]))
        else:
            ts.append(time.perf_counter() - t0)
    kname = e.kernel_name()
    e.close()
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": round(n_streams * T / SR / med, 1), "ms_per_step": round(med / steps * 1e3, 6), "kernel": kname,
            "seconds_per_stream": round(T / SR, 2), "steps_per_pass": steps, "passes": passes, "max_abs_err_vs_oracle": err,
            "checked": "stream 0 = input.wav looped, all 10 s; last stream, 2,560 frames",
            "note": "SURVEY 8(d) input recipe (stream 0 = example_audio/input.wav looped, the others two-tones at (1 + s/n) x 220 / 1230 Hz, "
                    "0.35 peak), median of the passes; a fidelity run beside `value`, not `value`"}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--reps", type=int, default=None,
                    help="timed regions of exactly --steps steps, each behind its own --warmup untimed steps; value = the "
                         "median region. Default: 11, more for short regions (about 30 ms of timed work, at most 101)")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config (see the module docstring)")
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (weak scaling); default from --config")
    ap.add_argument("--block", type=int, default=64)
    ap.add_argument("--model", default=None, help="fixture name under tests/golden/models; default from --config")
    ap.add_argument("--fast-tanh", type=int, default=1, help="benchmodel default: fast tanh ON (tools/benchmodel.cpp:27)")
    ap.add_argument("--launch", choices=["block", "resident"], default="block")
    ap.add_argument("--kernel", choices=["auto", "generic", "a1", "a1_mfma", "a1_il", "wn_reg"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-runs", action="store_true", help="skip the latency pass and the fast_tanh-off / zeros-input runs")
    ap.add_argument("--slim-mix", action="store_true",
                    help="slimmable models: stream s runs at ratio (0.0, 0.34, 0.67, 1.0)[s %% 4] (BASELINE.json configs[4])")
    ap.add_argument("--pre-reps", type=int, default=None,
                    help="untimed repetitions (W + K steps each, same calls) in front of the timed ones; default: ~80 ms worth")
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="untimed GPU work on a scratch batch before the warm-up steps (brings the clocks up; the measured "
                         "batch's state is not touched)")
    ap.add_argument("--check", type=int, default=1, help="verify stream 0 of rank 0 against the oracle after timing")
    ap.add_argument("--persistent", type=int, default=1,
                    help="--launch block: use the persistent block mode when the batch is eligible (one resident launch, one "
                         "doorbell per step); 0 = one kernel launch per step")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the distributed branch (RCCL process group, model broadcast, scatter / gather, barriers, all_reduce) even "
                         "with WORLD_SIZE=1: the only way to execute it on a one-GPU box (tests/test_gpu_parity.py)")
    ap.add_argument("--dry-run", action="store_true", help="CPU tensors + gloo + a stub instead of the kernels (plumbing test)")
    ap.add_argument("--brief", action="store_true",
                    help="a short run for the `other_configs` block of the default invocation: no latency pass / side runs / "
                         "resident comparison, a ~2 s CPU baseline")
    ap.add_argument("--full-line", action="store_true",
                    help="print the whole record as the line (default: a compact line of at most 6 KB, the whole record goes to "
                         "gpurun_out/bench_full.json)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default invocation (config 2, one GPU): do not append brief runs of configs 3, 4 and 5")
    args = ap.parse_args()
    if args.brief:
        args.no_side_runs = True
    cfg = CONFIGS[args.config]
    model_name = args.model or cfg["model"]
    n_streams = args.streams or cfg["streams"]
    slim_mix = args.slim_mix or (cfg["slim_mix"] and args.model is None)

    import numpy as np
    import torch
    import torch.distributed as dist
    from signals import stream_bank
    from neuralampmodelercore_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_distributed
    if distributed and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not distributed and args.gpus != 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if args.dry_run:
        dev = torch.device("cpu")
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=dev)

    model_path = os.path.join(ROOT, "tests", "golden", "models", model_name + ".nam")
    block, K, W = args.block, args.steps, args.warmup
    # repetitions: an odd count; short regions (the driver's --steps 20 is 0.3 ms) get more of them
    R = max(1, args.reps) if args.reps is not None else min(101, max(11, int(30e-3 / max(K * 15e-6, 1e-9)) | 1))
    T = (W + K) * block

    # ---- one-time weight broadcast: rank 0 reads the file, the text travels once over RCCL ----
    model_text = None
    if distributed:
        model_text = sharding.broadcast_model_text(model_path if rank == 0 else None, src=0, device=dev)
    model = nam = None
    ic = oc = 1
    if not args.dry_run:
        import neuralampmodelercore_amd as nam
        model = nam.get_dsp_json(model_text, fast_tanh=bool(args.fast_tanh)) if distributed else nam.get_dsp(
            model_path, fast_tanh=bool(args.fast_tanh))
        ic, oc = model.NumInputChannels(), model.NumOutputChannels()

    # ---- ownership: global stream s has width class s % 4 when widths are mixed; every rank gets the same mix ----
    n_total = n_streams * world
    classes_global = [s % len(SLIM_RATIOS) for s in range(n_total)] if slim_mix else [0] * n_total
    owners = [sharding.shard_by_class(classes_global, r, world) for r in range(world)]
    mine = owners[rank]
    n_local = len(mine)  # == n_streams unless a width class does not divide by the world size (ragged shards are fine)
    local_classes = [classes_global[s] for s in mine] if slim_mix else None

    # ---- synthetic input bank: generated on rank 0, scattered to the ranks over RCCL ----
    bank = stream_bank(n_total, T, seed=0) if rank == 0 else None  # [world * n_streams, T]
    if distributed:
        full = torch.from_numpy(bank[:, None, :]).to(dev) if rank == 0 else None
        x = sharding.scatter_rows(full, owners, src=0, device=dev)
        del full
    else:
        x = torch.from_numpy(bank[:, None, :]).to(dev)
    if ic > 1:  # multi-channel models: every input channel carries the stream's signal
        x = x.repeat(1, ic, 1).contiguous()
    assert tuple(x.shape) == (n_local, ic, T)
    y = torch.zeros((n_local, oc, T), dtype=torch.float32, device=dev)

    def fence(engine):
        """barrier + device synchronize: this rank's work is done (synchronize), then every rank's is (barrier), then the
        barrier's own collective has left the device (synchronize). Returns the host clock right behind the FIRST synchronize:
        a timed region ends when this rank's K steps are done — the MAX over ranks of those is when the whole job's are; the
        barrier that follows brackets the region (nothing of the next one starts before every rank is here) but its own
        collective (a launch + an all-reduce over xGMI: tens of microseconds, a third of a 20-step region) is not a step."""
        engine.sync()
        t_done = time.perf_counter()
        if distributed:
            dist.barrier()
            engine.sync()
        return t_done

    def reduce_max(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    scratch = None
    if args.dry_run:
        engine = StubEngine(torch, block)
    else:
        if args.spinup_ms > 0:  # (launch per step / resident launches; never the persistent mode)
            scratch = HipEngine(nam, torch, model, n_local, block, local_rank, args.kernel, local_classes)
        engine = HipEngine(nam, torch, model, n_local, block, local_rank, args.kernel, local_classes,
                           persistent=bool(args.persistent) and args.launch == "block")
    engine.bind(x, y)
    if not args.dry_run and scratch is not None:
        # Spin-up, right in front of the repetitions: continuous untimed GPU work on a SCRATCH batch (the measured batch's
        # state is not touched; the output window is zeroed again afterwards). The device takes tens of milliseconds of
        # load to reach its sustained clocks, and the driver-shaped run (--steps 20 --warmup 5: 0.3 ms per repetition)
        # would otherwise be measured entirely inside that ramp (docs/DESIGN_HISTORY.md §6: the same 20-step region
        # 246 us at the start of a process, 226 us 20 ms later).
        scratch.bind(x, y)
        t_end = time.perf_counter() + args.spinup_ms / 1e3
        while time.perf_counter() < t_end:
            for _ in range(8):
                scratch.run_steps(0, min(W + K, 64), "resident")
            scratch.sync()
        y.zero_()
        engine.sync()

    # ---- R times: W untimed warm-up steps, then a timed region of exactly K steps between barrier + synchronize pairs
    # (warm-up walks blocks 0..W of the window, the region blocks W..W+K; the streams' state runs on across repetitions).
    # All bookkeeping (max over ranks) happens after the last repetition, so that nothing but the fence sits between a
    # repetition's warm-up and its timed steps.
    raw = []
    got_dev = None
    n_chk = min(T, 64 * 40)
    # EVERY rank checks itself (each GPU renders its own shard, NAM/wavenet/model.cpp:822-910 per stream): the first
    # and the last stream it owns, against the oracle, after timing; the line carries the MAX over ranks and every rank's own figure
    chk_rows = sorted({0, n_local - 1}) if n_local > 0 else []
    pers = getattr(engine, "persistent", False)
    # In front of them P untimed repetitions of exactly the same shape: the device's clocks settle over tens of milliseconds
    # of THIS duty cycle (regions of the driver's 20-step shape run 9.0 -> 9.5 -> 8.3 us per step over the first 60 ms,
    # profiles/r03/bench_default_driver_shape.json: `repetitions`), which the spin-up on the scratch batch (one resident
    # launch after another) does not reproduce; the R timed repetitions then sit in the sustained state.
    P = max(0, args.pre_reps) if args.pre_reps is not None else (0 if args.dry_run else min(400, int(80e-3 / max((W + K) * 10e-6, 1e-9))))
    for rep in range(-P, R):
        if rep < 0:
            engine.run_steps(0, W, args.launch)
            fence(engine)
            engine.run_steps(W, K, args.launch)
            fence(engine)
            if rep == -P:
                # parity sample: the first buffers of the FIRST pass over the window. The copy runs on torch's stream, the next
                # repetition's session launch on its own: WAIT for the copy, or a slow first launch of the copy kernel (module
                # load on some boxes) lets the second pass overwrite what it is about to read (round 5: three runs in seventy
                # read max-abs 0.562 = exactly |first pass - second pass| of the checked stream)
                got_dev = y[chk_rows, 0, :n_chk].clone()
                if not args.dry_run:
                    torch.cuda.synchronize(dev)
            continue
        engine.run_steps(0, W, args.launch)
        fence(engine)
        t0 = time.perf_counter()
        e0 = None if pers else engine.event()
        engine.run_steps(W, K, args.launch)
        t_enq = time.perf_counter() - t0
        e1 = None if pers else engine.event()
        wall = fence(engine) - t0
        # the K steps' own time: HIP events on the launch stream around the K launches — or, in persistent block mode
        # (the work runs on the session's stream, not between two events of the caller's), the host clock from before
        # the first command until the host has seen every workgroup publish the last buffer (launch latency included)
        gpu_s = (engine.t_flushed - t0) if pers else engine.elapsed_ms(e0, e1) / 1e3
        raw.append((wall, gpu_s, t_enq))
        if rep == 0 and P == 0:
            got_dev = y[chk_rows, 0, :n_chk].clone()  # parity sample: blocks 0..W+K of the FIRST pass over the window
            if not args.dry_run:
                torch.cuda.synchronize(dev)  # (see above: the copy must not race the next repetition)
    red = reduce_max([v for r_ in raw for v in r_[:2]])
    regions = [{"wall_s": red[2 * i], "gpu_s": red[2 * i + 1], "enqueue_s": raw[i][2],
                "flushed_s": raw[i][1] if pers else None} for i in range(R)]
    # the kernel the timed regions ran (asked now: a session that sees bursts of a few buffers — the latency pass below — starts
    # the low-latency form of the headline kernel from then on, include/nam_hip.h: nam_hip_batch_set_persistent)
    kname_timed = engine.kernel_name()
    order = sorted(range(R), key=lambda i: regions[i]["wall_s"])
    med = regions[order[R // 2]]
    wall_med, gpu_s_med = med["wall_s"], med["gpu_s"]

    # alongside (not `value`): the same K blocks as ONE resident launch — the offline re-amp shape, no per-block kernel
    # boundary — timed the same way
    other = None
    if args.launch == "block" and not args.brief:
        fence(engine)
        t1 = time.perf_counter()
        engine.run_steps(W, K, "resident")
        other = reduce_max([fence(engine) - t1])[0]

    # per-launch latency distribution (min / p50 / p99 / p99.9, tools/bench_a2_fast.cpp:274-296): a separate pass with a
    # HIP event between launches on the launch stream (the events cost a little, so this is never the timed region)
    latency = None
    if rank == 0 and not args.no_side_runs and args.launch == "block":
        n_lat = min(K, 1000)
        fence_local = engine.sync
        fence_local()
        if getattr(engine, "persistent", False):
            # persistent block mode: what a real-time host sees per buffer — ring the doorbell, wait until every workgroup
            # has published the buffer (the session stays alive between buffers)
            for _ in range(8):  # (untimed: the session learns the pattern — one-buffer bursts — before the samples are taken)
                engine.run_steps(W, 1, "block")
                engine.batch.flush(engine.stream.cuda_stream)
            d = []
            for s in range(n_lat):
                t_a = time.perf_counter()
                engine.run_steps(W + s, 1, "block")
                engine.batch.flush(engine.stream.cuda_stream)
                d.append((time.perf_counter() - t_a) * 1e6)
            d.sort()
            note = ("us per buffer, host clock: doorbell -> all workgroups done -> host (persistent block mode, session alive); "
                    "kernel of these one-buffer bursts: " + engine.kernel_name())
        else:
            evs = [engine.event()]
            for s in range(n_lat):
                engine.run_steps(W + s, 1, "block")
                evs.append(engine.event())
            fence_local()
            d = sorted(engine.elapsed_ms(evs[i], evs[i + 1]) * 1e3 for i in range(n_lat))
            note = "us between consecutive HIP events on the launch stream, one launch per step (event overhead included)"
        fence_local()
        latency = {"launches": n_lat, "kernel": engine.kernel_name(), "min": round(d[0], 2), "p50": round(percentile(d, 0.5), 2),
                   "p99": round(percentile(d, 0.99), 2), "p99_9": round(percentile(d, 0.999), 2), "max": round(d[-1], 2),
                   "note": note}
    if distributed:
        dist.barrier()

    # after the timed regions: gather the last rendered block of every stream back to rank 0 over RCCL
    tail = y[:, :, (W + K - 1) * block:].contiguous()
    gathered = sharding.gather_rows(tail, owners, n_total, dst=0) if distributed else tail
    finite = bool(torch.isfinite(y).all()) and (gathered is None or bool(torch.isfinite(gathered).all()))
    gathered_ok = gathered is None or tuple(gathered.shape) == (n_total, oc, block)

    parity = None
    parity_detail = None
    parity_per_rank = None
    parity_streams = [int(mine[i]) for i in chk_rows]  # (global stream numbers this rank checked)
    if args.check and chk_rows:
        if args.dry_run:
            parity = float(torch.max(torch.abs(got_dev - 0.5 * x[chk_rows, 0, :n_chk])))
        else:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import nam_oracle
            got_np = got_dev.cpu().numpy()
            sig_np = x[chk_rows, :, :n_chk].cpu().numpy()  # this rank's OWN rows of the scattered bank (every input channel)
            parity = 0.0
            for j, i in enumerate(chk_rows):
                ref = nam_oracle.get_dsp(model_path, fast_tanh=bool(args.fast_tanh))
                if slim_mix:
                    ref.SetSlimmableSize(SLIM_RATIOS[classes_global[mine[i]]])
                ref.Reset(SR, block)
                r = ref.process_stream(sig_np[j] if ic > 1 else sig_np[j, 0], block)[0]
                err = float(np.max(np.abs(r - got_np[j])))
                parity = max(parity, err)
                if err > 1e-3 and parity_detail is None:
                    # a wrong result is a finding, not a number: say where (which frames of the checked stream, whole buffers of zeros?)
                    bad = np.nonzero(np.abs(r - got_np[j]) > 1e-3)[0]
                    blocks = sorted(set((bad // block).tolist()))
                    parity_detail = {"rank": rank, "stream": int(mine[i]), "bad_frames": int(bad.size), "first": int(bad[0]), "last": int(bad[-1]),
                                     "blocks": blocks[:16], "n_blocks": len(blocks),
                                     "zero_blocks": [int(b_) for b_ in blocks[:16] if not np.any(got_np[j, b_ * block:(b_ + 1) * block])]}
                    print("bench.py: PARITY FAILURE " + json.dumps(parity_detail), file=sys.stderr, flush=True)
    if distributed:
        # every rank's own figure travels to rank 0 (all_gather of one double per rank; -1 = that rank checked nothing)
        mine_t = torch.tensor([parity if parity is not None else -1.0], dtype=torch.float64, device=dev)
        all_t = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(all_t, mine_t)
        parity_per_rank = [float(t_[0]) for t_ in all_t]
        checked = [v for v in parity_per_rank if v >= 0.0]
        parity = max(checked) if checked else None
        streams_t = torch.tensor((parity_streams + [-1, -1])[:2], dtype=torch.int64, device=dev)
        all_s = [torch.zeros_like(streams_t) for _ in range(world)]
        dist.all_gather(all_s, streams_t)
        parity_streams = sorted({int(v) for t_ in all_s for v in t_ if int(v) >= 0})
    elif parity is not None:
        parity_per_rank = [parity]

    # side runs on rank 0 (N = 1): fast_tanh OFF and an all-zeros input (benchmodel's own input, tools/benchmodel.cpp:103-132)
    side = {}
    if rank == 0 and world == 1 and not args.no_side_runs and not args.dry_run:
        def oracle_err(fast_tanh_flag, sig_rows, got_rows, row_ids):
            """max |oracle - got| over the given local streams (fresh oracle each: Reset + prewarm, the signal from its first frame)"""
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import nam_oracle
            worst = 0.0
            for j, i in enumerate(row_ids):
                ref2 = nam_oracle.get_dsp(model_path, fast_tanh=fast_tanh_flag)
                if slim_mix:
                    ref2.SetSlimmableSize(SLIM_RATIOS[classes_global[mine[i]]])
                ref2.Reset(SR, block)
                r2 = ref2.process_stream(sig_rows[j] if ic > 1 else sig_rows[j, 0], block)[0]
                worst = max(worst, float(np.max(np.abs(r2 - got_rows[j]))))
            return worst

        def quick(engine2, xin, check_fast_tanh=None):
            # the headline's own protocol on another engine / input: the same launch mode (persistent block mode when the
            # headline runs it), warm-up steps, fence, a region of exactly K steps, fence; untimed regions of the same
            # shape in front (clocks settle under this duty cycle), the median of the timed ones
            engine2.bind(xin, y)
            err = None
            if check_fast_tanh is not None and args.check:
                # a fresh engine (Reset + prewarm, nothing run yet): its first pass over the window against the oracle with the
                # same activation setting and the same input — the session-mode instantiation of the kernel with libm tanh, and
                # benchmodel's own all-zeros input (tools/benchmodel.cpp:103-132), are CHECKED, not only timed
                engine2.run_steps(0, W + K, args.launch)
                engine2.sync()
                err = oracle_err(check_fast_tanh, xin[chk_rows, :, :n_chk].cpu().numpy(), y[chk_rows, 0, :n_chk].cpu().numpy(), chk_rows)
            ts = []
            n_pre, n_rep = min(P, 60), min(R, 21)
            for rep in range(-n_pre, n_rep):
                engine2.run_steps(0, W, args.launch)
                engine2.sync()
                t0 = time.perf_counter()
                engine2.run_steps(W, K, args.launch)
                engine2.sync()
                if rep >= 0:
                    ts.append(time.perf_counter() - t0)
            ts.sort()
            return {"value": round(n_streams * block * K / SR / ts[len(ts) // 2], 1), "ms_per_step": round(ts[len(ts) // 2] / K * 1e3, 6),
                    "kernel": engine2.kernel_name(), "persistent_block_mode": bool(getattr(engine2, "persistent", False)),
                    "regions": len(ts), "untimed_in_front": n_pre, "max_abs_err_vs_oracle": err}

        def fresh_engine(m_):
            return HipEngine(nam, torch, m_, n_streams, block, local_rank, args.kernel, local_classes,
                             persistent=bool(args.persistent) and args.launch == "block")
        e0 = fresh_engine(model)
        side["zeros_input"] = quick(e0, torch.zeros_like(x), check_fast_tanh=bool(args.fast_tanh))
        e0.close()
        m2 = nam.get_dsp(model_path, fast_tanh=not bool(args.fast_tanh))
        e2 = fresh_engine(m2)
        side["fast_tanh_off" if args.fast_tanh else "fast_tanh_on"] = quick(e2, x, check_fast_tanh=not bool(args.fast_tanh))
        e2.close()
        # SURVEY 8(d)'s input recipe as a fidelity run (never `value`; timing is data-independent): stream 0 = example_audio/input.wav
        # looped, stream s = the two-tone at (1 + s / n) x the frequencies and 0.35 peak, >= 10 s of audio per stream after Reset +
        # prewarm, 10 passes, the median (protocol of tools/benchmark_wavenet_a1.sh:10,75-95), buffer = 64 in the headline's launch mode
        if args.launch == "block" and ic == 1:
            try:
                side["wav_input"] = wav_fidelity_run(fresh_engine, model, n_streams, block, dev, oracle_err, bool(args.fast_tanh), torch, np)
            except Exception as e:  # noqa: BLE001  (a side figure must not take the line with it)
                side["wav_input"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        macs = model_macs(model_path)
        flops_per_sample = 2 * macs
        samples_per_step_gpu = n_streams * block
        total_samples = n_total * block * K
        xrt = total_samples / SR / wall_med
        launches = K if args.launch == "block" else 1
        avg_launch_s = gpu_s_med / launches
        samples_per_launch = samples_per_step_gpu * (1 if args.launch == "block" else K)
        flops_per_launch = flops_per_sample * samples_per_launch
        achieved_tf = flops_per_launch / avg_launch_s / 1e12
        mj = active_model_json(model_path)
        hist = wavenet_history_bytes_per_sample(mj["config"]) if mj["architecture"] == "WaveNet" else 0
        bytes_per_sample = hist + 4 * (ic + oc)
        achieved_gbs = bytes_per_sample * samples_per_launch / avg_launch_s / 1e9
        kname = kname_timed
        tr = measured_traffic(kname, model_name, n_streams, block, args.launch)
        lds = None
        if tr and tr.get("lds_idx_active_cycles") is not None and tr.get("rocprof_avg_launch_us"):
            # SQ_LDS_IDX_ACTIVE = all LDS-array cycles of the dispatch, summed over the CUs; SQ_LDS_BANK_CONFLICT = the extra
            # ones (MI355X_MICROARCH.md, LDS). Normalised by the PROFILED dispatch's own duration x the 2.4 GHz clock x 256
            # CUs (GRBM_GUI_ACTIVE is summed over the XCDs and is not a per-CU cycle count: round 2 divided by it and
            # printed a figure ten times too small). An LDS-array cycle moves at most 256 B, so the same counter bounds
            # the bytes per second: against the guide's ~150 TB/s for wide ds_reads chip-wide.
            cycles_avail = 256.0 * tr["rocprof_avg_launch_us"] * 1e-6 * LDS_CLOCK_HZ
            busy = tr["lds_idx_active_cycles"] / cycles_avail
            tbs = tr["lds_idx_active_cycles"] * 256.0 / (tr["rocprof_avg_launch_us"] * 1e-6) / 1e12
            lds = {"achieved": round(tbs, 2), "peak": LDS_PEAK_TBS, "unit": "TB/s (upper bound: LDS-array cycles x 256 B per profiled dispatch)",
                   "frac": round(tbs / LDS_PEAK_TBS, 4),
                   "array_busy_frac": round(busy, 4),
                   "bank_conflict_frac_of_active": round(tr.get("lds_bank_conflict_cycles", 0) / max(tr["lds_idx_active_cycles"], 1), 4),
                   "note": tr.get("lds_note", "rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT, per launch")
                           + f"; normalised by the profiled dispatch ({tr['rocprof_avg_launch_us']} us) x 2.4 GHz x 256 CUs"}
        # fractions against the WALL ms_per_step the line reports (round 2 divided by the host-visible time next to it)
        wall_launch_s = wall_med / launches
        contract_gbs_wall = bytes_per_sample * samples_per_launch / wall_launch_s / 1e9
        traffic_b = tr["hbm_bytes_per_launch"] if tr else None
        counter_frac = None if traffic_b is None else round(traffic_b / wall_launch_s / 1e9 / HBM_PEAK_GBS, 4)
        # The honest fraction: the physical floor of one step over the wall time of one step. Floor = the largest of (a) the
        # PMC-measured HBM bytes at the achievable 6.3 TB/s, (b) the algorithmic flops at the fp32 peak, (c) the cycles the
        # kernel's own instruction stream needs on the matrix / vector issue port of the busiest-on-average SIMD: every fp32
        # MFMA occupies the port for its pass count (SQ_VALU_MFMA_BUSY_CYCLES), every other vector instruction for one
        # quad-cycle, and the two never overlap on this chip (SQ_VALU_MFMA_COEXEC_CYCLES = 0 in every profile of this repo).
        floor_parts = None
        issue_cpi = None
        if tr is not None:
            floor_parts = {"hbm_us": traffic_b / (HBM_ACHIEVABLE_GBS * 1e9) * 1e6, "fp32_us": flops_per_launch / (FP32_PEAK_TFLOPS * 1e12) * 1e6}
            if tr.get("mfma_busy_cycles") is not None and tr.get("insts_per_launch", {}).get("VALU") is not None:
                # (kernels without matrix instructions too — nam_wn_reg_kernel: mfma_busy_cycles = 0 and the whole floor is the
                # vector instructions of a SIMD's lone wave at the measured one-wave issue rate; `simds_busy`: the SIMDs that
                # hold a wave at all — 768 lone waves of config 5 sit on 768 of the 1,024)
                n_simd = float(tr.get("simds_busy", 1024))
                other_valu = tr["insts_per_launch"]["VALU"] - tr.get("mfma_insts", 0.0)
                # cycles of the issue port per non-matrix vector instruction of a SIMD: MEASURED (tools/src/valu_rate.hip at the
                # kernel's waves per SIMD and instruction mix, profiles/r05/valu_rate_microbench.txt), recorded with the entry
                issue_cpi = float(tr.get("issue_cycles_per_valu_inst", ISSUE_CPI_DEFAULT))
                floor_parts["issue_us"] = (tr["mfma_busy_cycles"] + issue_cpi * other_valu) / n_simd / LDS_CLOCK_HZ * 1e6
        floor_us = None if floor_parts is None else round(max(floor_parts.values()), 3)
        # Which roofline bounds the kernel = the largest part of the floor (HBM bytes / flops / issue port). The contract's
        # vocabulary is "hbm" | "mfma": the flops and the issue port are both the matrix / vector pipe (fp32 MFMA peak == fp32
        # vector peak, MI355X_MICROARCH.md), `bound_detail` says which. Without a PMC pass for this kernel: flops against the
        # I/O bytes (8 B per stream-sample) decide.
        io_floor_us = 4 * (ic + oc) * samples_per_launch / (HBM_ACHIEVABLE_GBS * 1e9) * 1e6
        parts_for_bound = floor_parts or {"hbm_us": io_floor_us, "fp32_us": flops_per_launch / (FP32_PEAK_TFLOPS * 1e12) * 1e6}
        bound_detail = max(parts_for_bound, key=parts_for_bound.get)[:-3]
        achieved_tf_wall = flops_per_launch / wall_launch_s / 1e12
        contract_frac = contract_gbs_wall / HBM_PEAK_GBS
        hbm_contract = {"achieved": round(contract_gbs_wall, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(contract_frac, 4) if contract_frac <= 1.0 else None,
                        "label": ("SURVEY 8(d-ii) contract bytes / wall time per step" if contract_frac <= 1.0 else
                                  "n/a (rings in LDS): the contract counts every tap read as memory traffic; above the HBM peak"),
                        "bytes_per_stream_sample": bytes_per_sample}
        if bound_detail == "hbm":
            head = {"bound": "hbm", "achieved": round(contract_gbs_wall, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(min(contract_frac, 1.0), 4)}
        else:
            head = {"bound": "mfma", "achieved": round(achieved_tf_wall, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved_tf_wall / FP32_PEAK_TFLOPS, 4)}
        roofline = dict(head)
        roofline.update({
            "bound_detail": bound_detail,
            "traffic": traffic_b,
            "kernel": kname,
            "floor_frac": None if floor_us is None else round(floor_us / (wall_launch_s * 1e6), 4),
            "floor_parts_us": None if floor_parts is None else {k_: round(v_, 3) for k_, v_ in floor_parts.items()},
            "floor_us": floor_us,
            "issue_cycles_per_valu_inst": issue_cpi if floor_parts and "issue_us" in floor_parts else None,
            "counter_hbm_frac": counter_frac,
            "source": (tr.get("source") if tr else None),
            "basis": "achieved = algorithmic work of one step (SURVEY 8d: 2 x MACs, or the contract bytes) / the WALL ms_per_step of this line; "
                     "floor_frac = max(floor_parts_us) / wall time per step: measured HBM bytes at 6.3 TB/s, algorithmic flops at 157.3 TFLOP/s, "
                     "(fp32 MFMA pipe cycles + issue_cycles_per_valu_inst x other vector instructions) per SIMD at 2.4 GHz — the issue "
                     "part counts matrix and vector instructions only (LDS, scalar, waits are not in it: a lower bound); counters from "
                     "the PMC passes in profiles/ (`source`)",
            "traffic_note": (tr["note"] if tr else "no PMC pass committed for this exact kernel / model / launch shape"),
            "note": f"algorithmic {flops_per_sample} FLOP and {bytes_per_sample} B ({hist} history + {4 * (ic + oc)} I/O) per stream-sample x "
                    f"{samples_per_launch} stream-samples per launch; avg launch {avg_launch_s * 1e6:.2f} us "
                    + ("= results visible to the host / steps (persistent block mode: one resident launch consumes one doorbell per "
                       "step; HIP events on the caller's stream would only bracket the doorbells)"
                       if getattr(engine, "persistent", False) else "from HIP events on the launch stream (median region)"),
            "hbm_contract": hbm_contract,
            "lds": lds,
            "compute": {"achieved": round(achieved_tf_wall, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved_tf_wall / FP32_PEAK_TFLOPS, 4), "kernel_time_frac": round(achieved_tf / FP32_PEAK_TFLOPS, 4)},
        })
        out = {
            "metric": f"real-time audio streams (xRT) at 48 kHz, {model_name}",
            "value": round(xrt, 1),
            "unit": "xRT (48 kHz real-time streams sustained)",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(wall_med * 1e3 / K, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{model_name}.nam, {n_streams} concurrent streams per GPU, buffer={block} samples, "
                            f"fast_tanh={'on' if args.fast_tanh else 'off'}"
                            + (", mixed widths (ratios 0.0/0.34/0.67/1.0 interleaved)" if slim_mix else "")
                            + (f" ({cfg['name']})" if args.model is None and args.streams is None else ""),
                "baseline_config": args.config,
                "streams_per_gpu": n_streams, "block": block, "launch": args.launch,
                "kernel": kname,
                "persistent_block_mode": bool(getattr(engine, "persistent", False)),
                "sharding": f"streams x{world}, contiguous per width class (no data-path collective)",
            },
            "repetitions": {"n": R, "untimed_in_front": P,
                            "statistic": "median region (each region = exactly `steps` steps between barrier+sync pairs, max over ranks); "
                                         "`untimed_in_front` repetitions of the same shape run first (clocks settle under this duty cycle)",
                            "ms_per_step_all": [round(r_["wall_s"] * 1e3 / K, 6) for r_ in regions],
                            "ms_per_step_min": round(regions[order[0]]["wall_s"] * 1e3 / K, 6),
                            "ms_per_step_max": round(regions[order[-1]]["wall_s"] * 1e3 / K, 6)},
            "host_enqueue_us_per_step": round(med["enqueue_s"] / K * 1e6, 2),
            "region_us": {"enqueued": round(med["enqueue_s"] * 1e6, 1),
                          "synchronized_and_flushed": None if med["flushed_s"] is None else round(med["flushed_s"] * 1e6, 1),
                          "end_of_region": round(med["wall_s"] * 1e6, 1)},
            "roofline": roofline,
            "gpu_ms_total": round(gpu_s_med * 1e3, 3),
            "latency_us": latency,
            "resident_launch": (None if other is None else {
                "value": round(n_total * block * K / SR / other, 1), "ms_per_step": round(other / K * 1e3, 6),
                "note": "same K blocks as one launch walking device-resident audio (offline re-amp shape); not `value`"}),
            "finite": finite and gathered_ok,
            "max_abs_err_vs_oracle": parity,
            "parity_per_rank": parity_per_rank,
            "parity_streams_checked": parity_streams,
        }
        if parity_detail is not None:
            out["parity_detail"] = parity_detail
        out.update(side)
        if args.dry_run:
            out["data"] = "dry-run (stub compute on CPU tensors over gloo): plumbing only, not a measurement"
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            out["cpu_baseline"] = cpu_baseline(model_path, bool(args.fast_tanh), block, target_seconds=2.0 if args.brief else 12.0,
                                               extras=not args.brief)
    # the default invocation (what the driver runs: config 2, one GPU) also carries BRIEF runs of configs 3, 4 and 5 — each
    # its own process with the same protocol (persistent block mode, median of the regions, roofline, parity against the
    # oracle, a ~2 s CPU baseline) — so that one driver-timed line holds every BASELINE.json configuration a GPU can run
    if (rank == 0 and world == 1 and not distributed and args.config == 2 and args.model is None and args.streams is None
            and not args.brief and not args.no_other_configs and not args.dry_run):
        engine.close()
        engine = None
        if scratch is not None:
            scratch.close()
            scratch = None
        out["other_configs"] = run_other_configs(args)
        try:
            out["host_io"] = host_io(model_path, n_streams, bool(args.fast_tanh))
        except Exception as e:  # (a side figure must not take the line with it)
            out["host_io"] = {"error": f"{type(e).__name__}: {e}"}
        st = out["other_configs"].get("2_steady") or {}
        out["steady_state"] = {"value": st.get("value"), "ms_per_step": st.get("ms_per_step"), "steps_per_region": 500,
                               "max_abs_err_vs_oracle": st.get("max_abs_err_vs_oracle"),
                               "floor_frac": (st.get("roofline") or {}).get("floor_frac"),
                               "compute_frac": ((st.get("roofline") or {}).get("compute") or {}).get("frac"),
                               "note": "the same kernel, streams and session mode in regions of 500 steps (other_configs['2_steady']); "
                                       "`value` above is the driver's 20-step region, a third of which is launch, completion, prologue "
                                       "and the first buffer's way through the pipeline"}
    if rank == 0:
        if args.full_line:
            print(json.dumps(out), flush=True)
        else:
            full_path = write_full_record(out) if world == 1 or rank == 0 else None
            print(json.dumps(compact_line(out, full_path)), flush=True)
    if engine is not None:
        engine.close()
    if scratch is not None:
        scratch.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
