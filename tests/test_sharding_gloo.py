"""N > 1 path on CPU: world_size-2 gloo run of the stream sharding helpers (the same code bench.py and a
multi-GPU render use with the nccl/RCCL backend). The per-shard compute is stood in for by the CPU
oracle — the GPU kernels themselves are covered by the -m gpu tests."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, model_path

torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, T, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from neuralampmodelercore_amd import sharding
    import nam_oracle
    from signals import stream_bank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.from_numpy(stream_bank(n_streams, T, seed=7)[:, None, :]) if rank == 0 else None
    local = sharding.scatter_streams(full, n_streams, src=0)
    s, e = sharding.shard_range(n_streams, rank, world)
    assert local.shape == (e - s, 1, T)
    outs = []
    for i in range(e - s):
        m = nam_oracle.get_dsp(model_path("wavenet"), fast_tanh=True)
        m.Reset(48000.0, 64)
        outs.append(m.process_stream(local[i, 0].numpy(), 64))
    y_local = torch.from_numpy(np.stack(outs)) if outs else torch.zeros((0, 1, T))
    y = sharding.gather_streams(y_local, n_streams, dst=0)
    if rank == 0:
        ref = []
        for i in range(n_streams):
            m = nam_oracle.get_dsp(model_path("wavenet"), fast_tanh=True)
            m.Reset(48000.0, 64)
            ref.append(m.process_stream(full[i, 0].numpy(), 64))
        q.put(float(np.max(np.abs(y.numpy() - np.stack(ref)))))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_exactly():
    from neuralampmodelercore_amd import sharding
    for n in (1, 7, 8, 256, 4096, 5):
        for w in (1, 2, 3, 8):
            rs = [sharding.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = sharding.shard_sizes(n, w)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(120)
def test_scatter_process_gather_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, 192, q)) for r in range(2)]
    for p in procs:
        p.start()
    err = q.get(timeout=100)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert err == 0.0
