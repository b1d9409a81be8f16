"""Frame-parallel sharding across ranks (one process per GPU) and timeline reassembly.

The path shards into independent units: per-frame work has no dependencies, per-pair work
(Farneback, Hamming) depends only on the previous sampled frame (flow is never warm-started,
reference app/analyzers/video.py:45).  So
  * many clips  -> whole clips per rank, no halo;
  * one clip    -> contiguous ranges of sampled frames per rank with a ONE-frame halo.
There is no data-path collective; the only exchange is one all-gather of the fixed-size
per-frame records (32 B each) to reassemble the timeline -- RCCL over xGMI on GPUs
(backend "nccl"), gloo in CPU tests.  The message is a few KB: latency-bound.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np

from ._lib import RECORD_DTYPE


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Balanced contiguous split of range(n_items): the first (n % world) ranks get one more."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_with_halo(n_frames: int, rank: int, world: int) -> Tuple[int, int, int]:
    """-> (load_start, start, end): the rank analyses frames [load_start, end) and keeps the
    records of [start, end); load_start = start-1 (the halo) except for the first shard."""
    start, end = shard_range(n_frames, rank, world)
    load_start = start - 1 if (start > 0 and end > start) else start
    return load_start, start, end


def analyze_shard(analyze_records: Callable[[np.ndarray], np.ndarray], frames, rank: int, world: int) -> np.ndarray:
    """Run ``analyze_records`` (frames -> records) on this rank's shard of ONE clip.
    ``frames`` may be the whole clip (indexable by slice); only the shard (+halo) is touched."""
    load_start, start, end = shard_with_halo(len(frames), rank, world)
    if end <= start:
        return np.zeros(0, RECORD_DTYPE)
    rec = analyze_records(frames[load_start:end])
    return rec[start - load_start:]


def all_gather_records(local: np.ndarray, group=None, device=None) -> np.ndarray:
    """One all-gather of variable-length record arrays, concatenated in rank order.
    Works with any initialised torch.distributed backend; ``device`` must be a cuda device
    for nccl (RCCL) and None / 'cpu' for gloo."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return local
    dev = torch.device(device) if device is not None else torch.device("cpu")
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts_h: List[int] = [int(c) for c in counts.cpu().tolist()]
    cap = max(max(counts_h), 1)
    itemsize = RECORD_DTYPE.itemsize
    send = torch.zeros(cap * itemsize, dtype=torch.uint8, device=dev)
    if len(local):
        raw = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).reshape(-1).copy())
        send[: raw.numel()] = raw.to(dev)
    recv = torch.empty(world * cap * itemsize, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    host = recv.cpu().numpy().reshape(world, cap * itemsize)
    parts = [host[r, : counts_h[r] * itemsize].copy().view(RECORD_DTYPE) for r in range(world)]
    return np.concatenate(parts) if parts else np.zeros(0, RECORD_DTYPE)


def gather_fixed(local: np.ndarray, group=None, device=None) -> np.ndarray:
    """All-gather when every rank holds the same number of records (whole clips per rank):
    a single collective, no count exchange."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return local
    dev = torch.device(device) if device is not None else torch.device("cpu")
    send = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).reshape(-1).copy()).to(dev)
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv.cpu().numpy().view(RECORD_DTYPE)
