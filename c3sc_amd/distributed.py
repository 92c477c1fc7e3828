"""Multi-GPU plumbing for the Bellman sweep: one process per GPU, fibers sharded in
contiguous blocks, NO collective in the data path; the only exchange steps are
  * the all-gather of the updated FT cores at the end of a value-iteration sweep (<= 357 KiB), and
  * (when a cross-approximation step needs every fiber on every rank) the all-gather of fiber outputs.
Backend "nccl" is RCCL over xGMI on the GPU box; the same code runs on "gloo" for the CPU tests.
Messages here are tiny-to-moderate, so a single all_gather_into_tensor per exchange (direct, one link per
peer on xGMI) is used rather than bucketed rings."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_range(F: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of ceil(F/world) fibers owned by `rank` (last ranks may be short/empty)."""
    per = (F + world - 1) // world
    lo = min(F, rank * per)
    return lo, min(F, lo + per)


def pack_cores(cores: Sequence[np.ndarray]) -> Tuple[np.ndarray, List[int]]:
    """Flatten FT cores (reference layout) into one float64 vector + offsets[d+1]."""
    offs = [0]
    for c in cores:
        offs.append(offs[-1] + int(c.size))
    return np.concatenate([np.asarray(c, dtype=np.float64).reshape(-1) for c in cores]), offs


def padded_len(n: int, world: int) -> int:
    return ((n + world - 1) // world) * world


def allgather_cores(shard_t, world: int, force: bool = False):
    """All-gather equal-size shards of the flattened cores; returns the full (padded) flat tensor."""
    import torch
    import torch.distributed as dist

    full = torch.empty(shard_t.numel() * world, dtype=shard_t.dtype, device=shard_t.device)
    if world == 1 and not force:  # force: go through the collective with a single rank too (rehearsal)
        full.copy_(shard_t)
    else:
        dist.all_gather_into_tensor(full, shard_t.contiguous())
    return full


def allgather_fiber_outputs(local_out_t, F: int, world: int):
    """Gather per-rank output blocks (rows lo:hi of shard_range) into the full (F, N) array on every rank."""
    import torch
    import torch.distributed as dist

    per = (F + world - 1) // world
    N = local_out_t.shape[1]
    buf = torch.zeros((per, N), dtype=local_out_t.dtype, device=local_out_t.device)
    buf[: local_out_t.shape[0]] = local_out_t
    if world == 1:
        return buf[:F]
    full = torch.empty((per * world, N), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(full, buf)
    return full[:F]
