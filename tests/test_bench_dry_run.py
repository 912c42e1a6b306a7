"""bench.py's distributed branch without GPUs: two processes over gloo run the SAME code the 8-GPU bench runs —
model-text broadcast from rank 0, per-width-class sharding, scatter of the input bank, warm-up + timed regions with
their barriers and all_reduce(MAX), gather of the rendered tail — with a stub in place of the kernels (--dry-run).
Plus the unit tests of the class-aware partition (every GPU gets the same width mix)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_by_class_gives_every_rank_the_same_mix():
    from neuralampmodelercore_amd import sharding
    for n, world in ((768, 8), (4096, 8), (10, 3), (7, 2), (4, 8)):
        classes = [s % 4 for s in range(n)]
        owners = [sharding.shard_by_class(classes, r, world) for r in range(world)]
        flat = sorted(s for o in owners for s in o)
        assert flat == list(range(n))  # every stream owned exactly once
        for c in range(4):
            per_rank = [sum(1 for s in o if classes[s] == c) for o in owners]
            assert max(per_rank) - min(per_rank) <= 1  # balanced within every class
        assert all(o == sorted(o) for o in owners)
    # one class = the plain contiguous partition
    assert [sharding.shard_by_class([0] * 10, r, 3) for r in range(3)] == [list(range(*sharding.shard_range(10, r, 3))) for r in range(3)]


@pytest.mark.timeout(180)
@pytest.mark.parametrize("config,streams", [(5, 6), (2, 5)])
def test_bench_distributed_branch_world2_gloo(config, streams):
    from neuralampmodelercore_amd import sharding
    import bench
    n_total = 2 * streams
    classes = [s % len(bench.SLIM_RATIOS) for s in range(n_total)] if bench.CONFIGS[config]["slim_mix"] else [0] * n_total
    owners = [sharding.shard_by_class(classes, r, 2) for r in range(2)]
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "4", "--warmup", "2",
             "--reps", "3", "--config", str(config), "--streams", str(streams)],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=150) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]  # only rank 0 prints the JSON line
    last = outs[0][0].strip().splitlines()[-1]
    assert len(last.encode()) < 6000  # the driver parses the line out of an 8 KB tail of stdout (round 4's 25 KB line was cut)
    line = json.loads(last)
    assert json.loads(json.dumps(line)) == line
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["finite"] is True and line["max_abs_err_vs_oracle"] == 0.0
    # EVERY rank checks its own first and last stream after timing; the line carries each rank's figure and their MAX
    assert line["parity_per_rank"] == [0.0, 0.0]
    first_last = sorted({o for r in range(2) for o in (owners[r][0], owners[r][-1])})
    assert line["parity_streams_checked"] == first_last
    assert line["repetitions"]["n"] == 3 and line["repetitions"]["ms_per_step_min"] <= line["repetitions"]["ms_per_step_max"]
    assert line["roofline"]["bound"] in ("hbm", "mfma") and line["roofline"]["frac"] is not None
    assert line["value"] > 0 and line["config"]["streams_per_gpu"] == streams
    assert line["data"].startswith("dry-run")


def test_bench_line_stays_under_the_drivers_tail():
    """The ONE line bench.py prints: every committed full record of a default run (the driver's command, one GPU: headline +
    other configurations + host_io + side runs — profiles/r0N/bench_default_driver_shape*.json) compacts to < 6 KB with the
    contract's keys in it; the whole record goes to gpurun_out/bench_full.json instead."""
    import glob
    sys.path.insert(0, ROOT)
    import bench
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*", "bench_default_driver_shape*.json")))
    assert recs
    for path in recs:
        with open(path) as f:
            full = json.load(f)
        if "other_configs" not in full or "full_record" in full:  # (a compact line committed as such: nothing to compact)
            assert len(json.dumps(full).encode()) < bench.LINE_LIMIT, path
            continue
        big = full
        line = bench.compact_line(full, "gpurun_out/bench_full.json")
        text = json.dumps(line)
        assert len(text.encode()) < bench.LINE_LIMIT, (path, len(text))
        assert json.loads(text) == line
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in line, (path, k)
        assert "workload" in line["config"] and set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
        assert set(line["other_configs"]) >= {"3", "4", "5"}
    # and a record blown up well past anything this file produces still comes out under the limit (optional blocks are shed)
    big["other_configs"] = {str(i): dict(big["other_configs"]["3"]) for i in range(400)}
    assert len(json.dumps(bench.compact_line(big, None)).encode()) < bench.LINE_LIMIT


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("config,streams", [(2, 64), (5, 96)])
def test_bench_distributed_branch_on_rccl_one_rank(config, streams):
    """The same distributed branch on the GPU with the REAL backend: launched the way the driver launches N > 1
    (python -m torch.distributed.run, one rank per GPU) but with one rank — RCCL process group bound to the device,
    model-text broadcast, scatter / gather of device tensors, barriers and all_reduce(MAX) around the timed regions,
    model.batch(device=LOCAL_RANK), persistent sessions per rank — and the kernels instead of the stub. A one-GPU box
    cannot do more; N = 2, 4, 8 are the driver's."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-distributed", "--steps", "20",
         "--warmup", "5", "--reps", "3", "--config", str(config), "--streams", str(streams), "--no-cpu-baseline",
         "--no-side-runs", "--spinup-ms", "0"],
        env=env, capture_output=True, text=True, timeout=550, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["finite"] is True
    assert line["max_abs_err_vs_oracle"] <= 5e-5 and line["value"] > 0
    assert line["parity_per_rank"] == [line["max_abs_err_vs_oracle"]] and len(line["parity_streams_checked"]) == 2
    assert line["config"]["streams_per_gpu"] == streams
