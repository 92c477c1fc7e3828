"""two gloo ranks on one GPU: sharded vs unsharded c3control_step_vi, sweep by sweep (debug of tests/test_distributed.py)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port):
    import torch.distributed as dist

    import facade_lib
    from c3sc_amd import workloads as wl
    from c3sc_amd.distributed import make_fiber_exchange

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = facade_lib.lib()
    for n in ("c3control_init_value", "c3control_step_vi", "valuef_copy"):
        getattr(L, n).restype = C.c_void_p
    L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
    L.valuef_get_cores.restype = C.POINTER(C.POINTER(C.c_double))
    L.valuef_norm2diff.restype = C.c_double
    L.valuef_norm.restype = C.c_double
    ex = make_fiber_exchange(world, rank)
    w = wl.c4_car7d().scaled(ngrid=(11,) * 7, rank=4)
    ctl = facade_lib.Control(w)
    aa = C.c_void_p(L.approx_args_init())
    L.approx_args_set_cross_tol(aa, C.c_double(1e-10))
    L.approx_args_set_round_tol(aa, C.c_double(1e-9))
    L.approx_args_set_kickrank(aa, C.c_size_t(2))
    L.approx_args_set_startrank(aa, C.c_size_t(3))
    L.approx_args_set_maxrank(aa, C.c_size_t(6))
    one = facade_lib.FIBER_FN(lambda n, x, out, a: (np.ctypeslib.as_array(out, shape=(n,)).fill(1.0), 0)[1])
    v0 = C.c_void_p(L.c3control_init_value(ctl.h, one, None, aa, 0))
    ne = C.c_size_t(0)
    state = v0
    for sweep in range(4):
        L.c3control_set_fiber_sharding(ctl.h, C.c_size_t(world), C.c_size_t(rank), ex, None)
        a = C.c_void_p(L.c3control_step_vi(ctl.h, state, aa, ctl.opt, 0, C.byref(ne)))
        na = ne.value
        L.c3control_set_fiber_sharding(ctl.h, C.c_size_t(1), C.c_size_t(0), None, None)
        b = C.c_void_p(L.c3control_step_vi(ctl.h, state, aa, ctl.opt, 0, C.byref(ne)))
        c = C.c_void_p(L.c3control_step_vi(ctl.h, state, aa, ctl.opt, 0, C.byref(ne)))
        ra = [int(L.valuef_get_ranks(a)[i]) for i in range(8)]
        rb = [int(L.valuef_get_ranks(b)[i]) for i in range(8)]
        print(f"rank {rank} sweep {sweep}: |sharded - unsharded| = {L.valuef_norm2diff(a, b):.3e}  |unsharded - unsharded again| = {L.valuef_norm2diff(b, c):.3e}  "
              f"|V| = {L.valuef_norm(b):.6e} ranks {ra} {rb} evals {na} {ne.value}", flush=True)
        state = b
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket

    import torch.multiprocessing as mp

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join()
