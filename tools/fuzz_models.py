"""Developer tool: random A1-family models (seeded) through every kernel that can run them, against the oracle.
Single-array models with random per-layer kernel sizes / dilations / head taps reach the K-tap MFMA kernel;
multi-array kernel-size-3 models reach the wave-specialised one; narrow ones (1 .. 4 channels, odd dilations up to 700)
the register-resident kernel's LDS rings, plain-layer runs and run-time-flag layers; every fourth model is FEATURE-RICH
(make_synthetic_models.random_featured: per-layer gating / blending, FiLM subsets with and without shift, grouped input /
mixin / 1x1 / head1x1, bottleneck != channels, a nested condition_dsp, PReLU / LeakyHardtanh / Hardswish secondaries —
alternately with free dimensions (op interpreter) and with the dimensions nam_wn_reg_kernel instantiates).
Every kernel runs 64-frame launches, one multi-block launch (the pipelined forms) and — where the batch is eligible — a
persistent session. Usage: python tools/fuzz_models.py [n] [seed] [--load-only]"""
import json, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import neuralampmodelercore_amd as nam
import nam_oracle
import make_synthetic_models as msm
from signals import stream_bank

ACTS = ["Tanh", "ReLU", "Sigmoid", "Hardtanh", "SiLU", "Softsign", "Hardswish", dict(type="LeakyReLU", negative_slope=0.03)]


def random_ktap(rng, tmp, idx):
    C = int(rng.choice([4, 8, 8, 12, 16]))
    n_layers = int(rng.integers(6, 14))
    ks = [int(rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 12, 13, 15, 16])) for _ in range(n_layers)]
    dl = [int(rng.choice([1, 1, 2, 3, 5, 8, 16, 31, 64, 65, 100, 200, 333])) for _ in range(n_layers)]
    act = ACTS[int(rng.integers(len(ACTS)))]
    spec = dict(C=C, ksizes=ks, dils=dl, act=act, head_k=int(rng.choice([1, 2, 6, 7, 16])), head_dil=int(rng.choice([1, 1, 2, 3])),
                head_bias=bool(rng.integers(2)), seed=int(rng.integers(1 << 30)))
    name = f"fuzz_kt_{idx}"
    old = msm.HERE
    msm.HERE = tmp
    os.makedirs(os.path.join(tmp, "models"), exist_ok=True)
    msm.build_ktap(name, **spec)
    msm.HERE = old
    return os.path.join(tmp, "models", name + ".nam"), spec


def random_multi(rng, tmp, idx):
    n_arr = int(rng.integers(2, 4))
    arrays = []
    for a in range(n_arr):
        C = int(rng.choice([4, 8, 12, 16, 5, 6, 7, 10, 14]))  # not a multiple of 4: zero-padded at plan time
        n_layers = int(rng.integers(3, 8))
        dl = [int(rng.choice([1, 2, 3, 4, 8, 16, 32, 33, 64, 100, 128, 256, 512])) for _ in range(n_layers)]
        arrays.append((C, dl, ["Tanh", "ReLU", "Sigmoid"][int(rng.integers(3))], bool(rng.integers(2))))
    name = f"fuzz_ws_{idx}"
    old = msm.HERE
    msm.HERE = tmp
    os.makedirs(os.path.join(tmp, "models"), exist_ok=True)
    msm.build(name, arrays, int(rng.integers(1 << 30)))
    msm.HERE = old
    return os.path.join(tmp, "models", name + ".nam"), dict(arrays=arrays)


def random_narrow(rng, tmp, idx):
    """1 .. 4 channels per array: nam_wn_reg_kernel (Tanh / ReLU layers fuse into runs, Sigmoid ones stay single layers)."""
    n_arr = int(rng.integers(1, 4))
    arrays = []
    for a in range(n_arr):
        C = int(rng.choice([1, 2, 3, 4]))
        n_layers = int(rng.integers(1, 7))
        dl = [int(rng.choice([1, 2, 3, 5, 7, 16, 31, 32, 33, 63, 64, 65, 100, 127, 128, 300, 512, 700])) for _ in range(n_layers)]
        arrays.append((C, dl, ["Tanh", "ReLU", "Sigmoid"][int(rng.integers(3))], bool(rng.integers(2))))
    name = f"fuzz_nw_{idx}"
    old = msm.HERE
    msm.HERE = tmp
    os.makedirs(os.path.join(tmp, "models"), exist_ok=True)
    msm.build(name, arrays, int(rng.integers(1 << 30)))
    msm.HERE = old
    return os.path.join(tmp, "models", name + ".nam"), dict(arrays=arrays)


def random_featured(rng, tmp, idx):
    """Feature-rich WaveNet (schema NAM/wavenet/model.cpp:913-1276): see make_synthetic_models.random_featured."""
    os.makedirs(os.path.join(tmp, "models"), exist_ok=True)
    path = os.path.join(tmp, "models", f"fuzz_ft_{idx}.nam")
    wr = bool((idx // 4) % 2)
    post = (idx // 8) % 2 == 1  # every other pair of feature-rich models: + a random post-stack head
    msm.write_featured(path, int(rng.integers(1 << 30)), wr_shapes=wr, post_head=post)
    return path, dict(featured=True, wr_shapes=wr, post_head=post)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    load_only = "--load-only" in sys.argv  # no GPU: just load every model (compiles the per-model kernels into the cache)
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(n):
            path, spec = (random_multi, random_ktap, random_narrow, random_featured)[i % 4](rng, tmp, i)
            ft = bool(rng.integers(2))
            model = nam.get_dsp(path, fast_tanh=ft)
            bits = model.info.has_a1_kernel
            if load_only:
                rng.integers(3, 7); rng.integers(0, 64)  # (keep the random sequence of a full run)
                print("loaded", os.path.basename(path), "bits", bits, flush=True)
                continue
            n_streams, block = 3, 64
            T = 64 * int(rng.integers(3, 7)) + int(rng.integers(0, 64))
            x = stream_bank(n_streams, T, seed=i)
            ic = model.NumInputChannels()
            if ic > 1:  # every input channel carries the stream's signal, scaled differently
                x = np.ascontiguousarray(np.stack([x * (1.0 - 0.3 * c) for c in range(ic)], axis=1))
            ref = nam_oracle.get_dsp(path, fast_tanh=ft)
            ref.Reset(48000.0, block)
            r = ref.process_stream(x[1], block)
            errs = {}
            kernels = [("generic", nam.KERNEL_GENERIC)] + ([("valu", nam.KERNEL_A1)] if bits & 1 else []) + ([("mfma", nam.KERNEL_A1_MFMA)] if bits & 2 else []) + ([("wn_reg", nam.KERNEL_WN_REG)] if bits & 16 else [])
            for kname, k in kernels:
                for mode, mf in (("blocks", block), ("one", 512), ("session", block)):
                    b = model.batch(n_streams, mf)
                    b.set_kernel(k)
                    if mode == "session" and not b.set_persistent(True):  # 64-frame buffers as commands of a persistent session
                        b.close()
                        continue
                    b.Reset(prewarm=True)
                    if mode == "one":
                        ref2 = nam_oracle.get_dsp(path, fast_tanh=ft); ref2.Reset(48000.0, mf); rr = ref2.process_stream(x[1], mf)
                    else:
                        rr = r
                    y = b.process_stream(x, mf)
                    errs[f"{kname}/{mode}"] = float(np.max(np.abs(y[1] - rr))) / max(1.0, float(np.max(np.abs(rr))))
                    b.close()
            worst = max(errs.values())
            ok = worst <= (5e-5 if ft else 1e-4) and np.isfinite(worst)
            bad += not ok
            print(("ok  " if ok else "FAIL"), os.path.basename(path), "bits", bits, "ft", int(ft), "T", T, "worst rel err %.2e" % worst,
                  "" if ok else (errs, spec), flush=True)
    print("FUZZ", "FAILED" if bad else "OK", f"({n} models)")


if __name__ == "__main__":
    main()
