"""Developer tool (profiles/r03/defit_table.txt): throughput of FiLM / gated / grouped models that are NOT
wavenet_a2_max on nam_wn_reg_kernel — compiled for the model's own shapes (the default) vs the run-time-flag
instantiations (NAM_HIP_JIT=0, where the model fits them) vs the op interpreter; 512 streams, persistent block mode where
the kernel has one, 64-frame buffers. xRT = streams * frames / 48000 / seconds."""
import os, sys, subprocess, json, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

WORKER = r'''
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, sys.argv[4]); sys.path.insert(0, os.path.join(sys.argv[4], "tests"))
import neuralampmodelercore_amd as nam
from signals import stream_bank
path, kernel, n_streams = sys.argv[1], sys.argv[2], int(sys.argv[3])
m = nam.get_dsp(path, fast_tanh=True)
ic = m.NumInputChannels()
b = m.batch(n_streams, 64)
b.set_kernel({"wn_reg": nam.KERNEL_WN_REG, "generic": nam.KERNEL_GENERIC}[kernel])
b.Reset(prewarm=True)
pers = b.set_persistent(True) if kernel == "wn_reg" else False
nb = 600
x = torch.from_numpy(np.repeat(stream_bank(n_streams, nb * 64, seed=1)[:, None, :], ic, axis=1).copy()).cuda()
y = torch.zeros((n_streams, m.NumOutputChannels(), nb * 64), device="cuda")
def run(k0, k1):
    for k in range(k0, k1):
        b.process_device(x.data_ptr() + k * 256, y.data_ptr() + k * 256, 64, nb * 64)
    b.flush(); torch.cuda.synchronize()
run(0, 100)
ts = []
for rep in range(5):
    t0 = time.perf_counter(); run(100, 600); ts.append(time.perf_counter() - t0)
ts.sort()
print(json.dumps({"kernel": b.kernel_name(), "persistent": bool(pers), "us_per_buffer": ts[2] / 500 * 1e6, "xrt": n_streams * 64 * 500 / 48000.0 / ts[2], "why": m.wr_why()}))
'''


def main():
    import make_synthetic_models as msm
    from conftest import model_path
    tmp = tempfile.mkdtemp()
    models = [("wavenet_a2_max (the tuned one)", model_path("wavenet_a2_max")), ("wavenet_condition_dsp", model_path("wavenet_condition_dsp")),
              ("synth_multich", model_path("synth_multich")), ("synth_leakyhardtanh", model_path("synth_leakyhardtanh"))]
    for seed in (1, 3, 4, 8, 10, 13):
        p = os.path.join(tmp, f"featured_{seed}.nam")
        msm.write_featured(p, 7000 + seed, wr_shapes=bool(seed % 2))
        models.append((f"featured seed {seed} ({'aot dims' if seed % 2 else 'free dims'})", p))
    print(f"{'model':38s} {'own shapes xRT':>15s} {'us/buf':>7s} | {'run-time flags':>15s} {'us/buf':>7s} | {'interpreter':>12s} {'us/buf':>7s}")
    for name, path in models:
        row = []
        for kernel, env in (("wn_reg", {"NAM_HIP_JIT": "1"}), ("wn_reg", {"NAM_HIP_JIT": "0"}), ("generic", {})):
            e = dict(os.environ); e.update(env)
            try:
                out = subprocess.run([sys.executable, "-c", WORKER, path, kernel, "512", ROOT], capture_output=True, text=True, timeout=300, env=e)
                line = [l for l in out.stdout.splitlines() if l.startswith("{")]
                j = json.loads(line[-1]) if line else None
            except Exception:
                j = None
            if j and (kernel == "generic" or j["kernel"] == "nam_wn_reg_kernel"):
                row.append(f"{j['xrt']:15.0f} {j['us_per_buffer']:7.2f}")
            else:
                row.append(f"{'—':>15s} {'':7s}")
        print(f"{name:38s} {row[0]} | {row[1]} | {row[2][3:]}", flush=True)


if __name__ == "__main__":
    main()
