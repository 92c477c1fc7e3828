"""N > 1 path on CPU: world_size-2 gloo.  The sharding / gather plumbing must reproduce the unsharded
result exactly.  (No GPU here, so the per-rank compute in this test is the oracle -- test infrastructure
standing in for the kernel; the plumbing under test is c3sc_amd/distributed.py.)"""
import os
import socket

import numpy as np
import pytest

from c3sc_amd import workloads as wl
from c3sc_amd.distributed import pack_cores, padded_len, shard_range


def test_shard_range_partitions():
    for F in (0, 1, 7, 100, 101, 1 << 17):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_range(F, world, r)
                assert 0 <= lo <= hi <= F
                seen += list(range(lo, hi))
            assert seen == list(range(F))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys

    import torch
    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib
    from c3sc_amd.distributed import allgather_cores, allgather_fiber_outputs

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = wl.c2_dubins().scaled(ngrid=(9, 8, 10), rank=3)
    cores = wl.synth_cores(w)
    flat, offs = pack_cores(cores)
    n = padded_len(len(flat), world)
    flat_p = np.concatenate([flat, np.zeros(n - len(flat))])
    shard = torch.from_numpy(flat_p.reshape(world, -1)[rank].copy())
    full = allgather_cores(shard, world).numpy()
    ok_cores = np.array_equal(full[: len(flat)], flat)
    # rebuild the cores from the gathered vector and run this rank's fiber block
    got = [full[offs[m]:offs[m + 1]].reshape(cores[m].shape) for m in range(w.dx)]
    P = oracle_lib.Problem(w, got)
    k, F = 1, 11
    idx = wl.synth_fibers(w, k, F)
    lo, hi = shard_range(F, world, rank)
    local, _, _ = P.bellman_fibers(k, idx[lo:hi]) if hi > lo else (np.zeros((0, w.ngrid[k])), None, None)
    allout = allgather_fiber_outputs(torch.from_numpy(local), F, world).numpy()
    if rank == 0:
        ref, _, _ = oracle_lib.Problem(w, cores).bellman_fibers(k, idx)
        q.put((ok_cores, bool(np.array_equal(allout, ref))))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_shard_and_gather(oracle):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_cores, ok_out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_cores and ok_out
