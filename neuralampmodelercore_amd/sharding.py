"""Stream sharding across the GPUs of one node (one process per GPU, torch.distributed).

Streams never interact (each has its own history rings / LSTM state; weights are read-only), so the
path shards with NO data-path collective: rank r of W owns the contiguous stream range
``shard_range(n, r, W)`` for the life of those streams. RCCL (backend "nccl" on ROCm; "gloo" in the
CPU tests) is only used to scatter input stream batches from the rank that read the audio and to
gather the rendered batches back — never inside the per-block hot path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple


def shard_range(n_streams: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first (n % world) ranks get one extra stream."""
    base, extra = divmod(n_streams, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_streams: int, world: int) -> List[int]:
    return [shard_range(n_streams, r, world)[1] - shard_range(n_streams, r, world)[0] for r in range(world)]


def shard_by_class(classes: Sequence[int], rank: int, world: int) -> List[int]:
    """Mixed batches (slimmable widths / container submodels): partition WITHIN each class so that every GPU gets the
    same class mix — stream s belongs to class classes[s]; returns the ascending global stream ids
    rank `rank` owns. Each class is split contiguously and balanced like shard_range; a class smaller than the world
    leaves some ranks without a member of it. Every stream is owned by exactly one rank."""
    members = {}
    for s, c in enumerate(classes):
        members.setdefault(c, []).append(s)
    mine: List[int] = []
    for c in sorted(members):
        ids = members[c]
        a, b = shard_range(len(ids), rank, world)
        mine.extend(ids[a:b])
    return sorted(mine)


def scatter_rows(full, owners: Sequence[Sequence[int]], src: int = 0, group=None, device=None):
    """Like scatter_streams for an arbitrary ownership: owners[r] = global stream ids of rank r (shard_by_class).
    Rank `src` holds `full` [n_streams, ch, T]; every rank returns its rows in the order of owners[rank]."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if rank == src:
        meta = torch.tensor([full.shape[1], full.shape[2]], dtype=torch.int64, device=full.device)
    else:
        meta = torch.zeros(2, dtype=torch.int64, device=device)
    dist.broadcast(meta, src=src, group=group)
    ch, T = int(meta[0]), int(meta[1])
    if rank == src:
        reqs = []
        for r in range(world):
            if r != src and len(owners[r]) > 0:
                idx = torch.as_tensor(list(owners[r]), dtype=torch.int64, device=full.device)
                reqs.append(dist.isend(full.index_select(0, idx).contiguous(), dst=r, group=group))
        idx = torch.as_tensor(list(owners[src]), dtype=torch.int64, device=full.device)
        local = full.index_select(0, idx).clone()
        for q in reqs:
            q.wait()
        return local
    local = torch.empty((len(owners[rank]), ch, T), dtype=torch.float32, device=device)
    if len(owners[rank]) > 0:
        dist.recv(local, src=src, group=group)
    return local


def gather_rows(local, owners: Sequence[Sequence[int]], n_streams: int, dst: int = 0, group=None):
    """Inverse of scatter_rows: rank `dst` returns [n_streams, ch, T] in global stream order, the others None."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if rank != dst:
        if local.shape[0] > 0:
            dist.send(local.contiguous(), dst=dst, group=group)
        return None
    full = torch.empty((n_streams,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        if len(owners[r]) == 0:
            continue
        idx = torch.as_tensor(list(owners[r]), dtype=torch.int64, device=local.device)
        if r == dst:
            full.index_copy_(0, idx, local)
        else:
            buf = torch.empty((len(owners[r]),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            dist.recv(buf, src=r, group=group)
            full.index_copy_(0, idx, buf)
    return full


def broadcast_model_text(path: Optional[str], src: int = 0, group=None, device=None) -> str:
    """One-time weight broadcast: rank `src` reads the .nam file, every rank receives its text (the
    weights travel once over RCCL instead of every rank parsing the file from shared storage) and loads it with
    get_dsp_json. a1_standard: 407 KB of JSON; the packed device plan is derived from it locally."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    if rank == src:
        with open(path, "rb") as f:
            data = f.read()
        n = torch.tensor([len(data)], dtype=torch.int64, device=device)
    else:
        data = b""
        n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(n, src=src, group=group)
    if rank == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device) if device is not None else torch.frombuffer(bytearray(data), dtype=torch.uint8)
    else:
        buf = torch.empty(int(n[0]), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src, group=group)
    return bytes(buf.cpu().numpy().tobytes()).decode("utf-8")


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
