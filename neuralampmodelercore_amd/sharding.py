"""Stream sharding across the GPUs of one node (one process per GPU, torch.distributed).

Streams never interact (each has its own history rings / LSTM state; weights are read-only), so the
path shards with NO data-path collective: rank r of W owns the contiguous stream range
``shard_range(n, r, W)`` for the life of those streams. RCCL (backend "nccl" on ROCm; "gloo" in the
CPU tests) is only used to scatter input stream batches from the rank that read the audio and to
gather the rendered batches back — never inside the per-block hot path.
"""
from __future__ import annotations

from typing import List, Optional, Tuple


def shard_range(n_streams: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first (n % world) ranks get one extra stream."""
    base, extra = divmod(n_streams, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_streams: int, world: int) -> List[int]:
    return [shard_range(n_streams, r, world)[1] - shard_range(n_streams, r, world)[0] for r in range(world)]


def scatter_streams(full, n_streams: int, src: int = 0, group=None, device=None):
    """Rank `src` holds `full` [n_streams, ch, T]; every rank gets its shard [n_local, ch, T].
    Implemented with point-to-point send/recv (shards may be ragged)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if rank == src:
        meta = torch.tensor([full.shape[1], full.shape[2]], dtype=torch.int64, device=full.device)
    else:
        meta = torch.zeros(2, dtype=torch.int64, device=device)
    dist.broadcast(meta, src=src, group=group)
    ch, T = int(meta[0]), int(meta[1])
    s, e = shard_range(n_streams, rank, world)
    if rank == src:
        reqs = []
        for r in range(world):
            rs, re = shard_range(n_streams, r, world)
            if r != src and re > rs:
                reqs.append(dist.isend(full[rs:re].contiguous(), dst=r, group=group))
        local = full[s:e].clone()
        for q in reqs:
            q.wait()
        return local
    local = torch.empty((e - s, ch, T), dtype=torch.float32, device=device)
    if e > s:
        dist.recv(local, src=src, group=group)
    return local


def gather_streams(local, n_streams: int, dst: int = 0, group=None):
    """Inverse of scatter_streams: rank `dst` returns [n_streams, ch, T], the others None."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if rank != dst:
        if local.shape[0] > 0:
            dist.send(local.contiguous(), dst=dst, group=group)
        return None
    full = torch.empty((n_streams,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        rs, re = shard_range(n_streams, r, world)
        if r == dst:
            full[rs:re] = local
        elif re > rs:
            dist.recv(full[rs:re], src=r, group=group)
    return full
