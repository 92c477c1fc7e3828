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


# the exchange callback of include/c3sc/valuefunc.h (c3sc_exchange_fn): int (*)(double *out, size_t F, size_t N, size_t lo, size_t hi, void *)
def make_fiber_exchange(world: int, rank: int, device=None, group=None):
    """All-gather of a core step's fiber values for libc3sc.so's sharded solver (c3control_set_fiber_sharding /
    valuef_interp_idx_sharded): rank r has filled rows [lo, hi) of the host array out[F][N]; afterwards every rank holds all F
    rows.  One all_gather_into_tensor of ceil(F/world) x N doubles per core step -- tens of KB, latency-bound -- through the
    process group's backend: RCCL over xGMI with a device staging tensor when `device` is given, gloo on host memory
    otherwise.  Returns the ctypes callback (keep a reference while the solver may call it)."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    EX = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p)

    def _exchange(out_p, F, N, lo, hi, _xarg):
        try:
            per = (F + world - 1) // world
            out = np.ctypeslib.as_array(out_p, shape=(F * N,)).reshape(F, N)
            buf = torch.zeros((per, N), dtype=torch.float64)
            if hi > lo:
                buf[: hi - lo] = torch.from_numpy(out[lo:hi])
            if device is not None:
                buf = buf.to(device)
            full = torch.empty((per * world, N), dtype=torch.float64, device=buf.device)
            dist.all_gather_into_tensor(full, buf, group=group)
            out[:] = full[:F].cpu().numpy()
            return 0
        except Exception as e:  # an exception must not unwind through the C frames
            print(f"c3sc fiber exchange failed on rank {rank}: {e!r}", flush=True)
            return 1

    return EX(_exchange)
