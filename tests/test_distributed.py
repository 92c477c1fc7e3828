"""N > 1 path on CPU: world_size-2 gloo.  The sharding / gather plumbing must reproduce the unsharded
result exactly.  (No GPU here, so the per-rank compute in this test is the oracle -- test infrastructure
standing in for the kernel; the plumbing under test is c3sc_amd/distributed.py.)"""
import os
import socket
import sys

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


# ---- the sharded solver path of libc3sc.so (valuef_interp_idx_sharded / c3control_set_fiber_sharding) ---------------------
def _sharded_worker(rank, world, port, q, use_gpu, mode="vi"):
    import ctypes as C
    import sys

    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import facade_lib
    import oracle_lib
    from c3sc_amd.distributed import make_fiber_exchange

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = facade_lib.lib()
    for n in ("valuef_interp_idx", "valuef_interp_idx_sharded", "c3control_init_value", "c3control_step_vi", "c3control_step_pi",
              "c3control_begin_pi", "valuef_copy"):
        getattr(L, n).restype = C.c_void_p
    L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
    L.valuef_get_cores.restype = C.POINTER(C.POINTER(C.c_double))
    ex = make_fiber_exchange(world, rank)

    def cores_of(vf, w):
        ranks = [int(L.valuef_get_ranks(vf)[i]) for i in range(w.dx + 1)]
        pp = L.valuef_get_cores(vf)
        return ranks, [np.ctypeslib.as_array(pp[m], shape=(w.ngrid[m] * ranks[m] * ranks[m + 1],)).copy() for m in range(w.dx)]

    def aargs():
        aa = C.c_void_p(L.approx_args_init())
        L.approx_args_set_cross_tol(aa, C.c_double(1e-10))
        L.approx_args_set_round_tol(aa, C.c_double(1e-9))
        L.approx_args_set_kickrank(aa, C.c_size_t(2))
        L.approx_args_set_startrank(aa, C.c_size_t(3))
        L.approx_args_set_maxrank(aa, C.c_size_t(6))
        return aa

    if not use_gpu:
        # CPU: the cross driver's sharded core steps with the oracle's bellman_vi as the fiber function (no device here)
        w = wl.c4_car7d().scaled(ngrid=(6, 5, 7, 5, 6, 5, 6), rank=3) if os.environ.get("C3SC_TEST_SHARD_7D") else wl.c2_dubins().scaled(ngrid=(9, 8, 10), rank=3)
        P = oracle_lib.Problem(w, wl.synth_cores(w))
        FI = C.CFUNCTYPE(C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p)
        calls = []

        polvf = None
        if mode == "pi":  # bellman_pi's fibers: the greedy policy of another value function, evaluated on P's (bellman.c:1702-1886)
            wp = wl.Workload(w.name, w.model, w.params, w.dx, w.du, w.lb, w.ub, w.ngrid, w.ranks, w.discount, w.bc, list(w.obstacles), w.cands)
            polvf = oracle_lib.ValueF(w.ngrid, w.ranks, [c * (1.0 + 0.3 * np.sin(np.arange(c.size)).reshape(c.shape)) for c in wl.synth_cores(wp)])

        def fi(F, k, idx_p, out_p, _a):
            idx = np.ctypeslib.as_array(idx_p, shape=(F, w.dx)).copy()
            out = np.ctypeslib.as_array(out_p, shape=(F, w.ngrid[k]))
            if mode == "fail" and rank == 1 and len(calls) >= 2:
                return 7  # this rank's fibers fail in its third core step: BOTH ranks must come back with an error, nobody may hang
            out[:] = P.policy_fibers(polvf, k, idx)[0] if mode == "pi" else P.bellman_fibers(k, idx, want_absorbed=False)[0]
            calls.append(F)
            return 0

        cb = FI(fi)
        xg = [facade_lib.f64(g) for g in w.xgrid()]
        gp, Ng, aa = facade_lib.ptrs(xg), facade_lib.usz(w.ngrid), aargs()
        a = C.c_void_p(L.valuef_interp_idx_sharded(C.c_size_t(w.dx), cb, None, facade_lib.sp(Ng), gp, None, aa, 0, C.c_size_t(world),
                                                   C.c_size_t(rank), ex, None))
        n_sharded = sum(calls)
        calls.clear()
        b = C.c_void_p(L.valuef_interp_idx(C.c_size_t(w.dx), cb, None, facade_lib.sp(Ng), gp, None, aa, 0))
        n_full = sum(calls)
        ra, ca = cores_of(a, w)
        rb, cb_ = cores_of(b, w)
        same = ra == rb and all(np.array_equal(x, y) for x, y in zip(ca, cb_))
        q.put((rank, same, n_sharded, n_full))
    else:
        # GPU: the whole solver step (c3control_step_vi, device kernels) sharded over two ranks on the one device of the box
        w = wl.c4_car7d().scaled(ngrid=(11,) * 7, rank=4)
        # value iteration: the reference's literal end-point rule (the rows other ranks computed enter this rank's memo, so even the
        # direction-dependent nodes agree); policy evaluation: the solver's default, consistent end points -- the per-node policy memo
        # holds this rank's rows only, and the bit-identity claim of include/c3sc/valuefunc.h is made for that rule
        ctl = facade_lib.Control(w, consistent_ends=(mode == "pi"))
        aa = aargs()
        zero = facade_lib.FIBER_FN(lambda n, x, out, a: (np.ctypeslib.as_array(out, shape=(n,)).fill(1.0), 0)[1])
        outs = []
        for sharded in (True, False):
            L.c3control_set_fiber_sharding(ctl.h, C.c_size_t(world if sharded else 1), C.c_size_t(rank), ex if sharded else None, None)
            vf = C.c_void_p(L.c3control_init_value(ctl.h, zero, None, aa, 0))
            ne = C.c_size_t(0)
            for _ in range(3 if mode == "vi" else 1):
                nxt = C.c_void_p(L.c3control_step_vi(ctl.h, vf, aa, ctl.opt, 0, C.byref(ne)))
                L.valuef_destroy(vf)
                vf = nxt
            if mode == "pi":  # c3control_step_pi x3 under the policy of the value function reached so far (bellman.c:2214-2262, 2343-2407)
                pol = C.c_void_p(L.c3control_begin_pi(ctl.h, vf))
                it = C.c_void_p(L.valuef_copy(vf))
                for _ in range(3):
                    nxt = C.c_void_p(L.c3control_step_pi(ctl.h, it, pol, aa, ctl.opt, 0, C.byref(ne)))
                    L.valuef_destroy(it)
                    it = nxt
                L.pi_param_destroy(pol)
                L.valuef_destroy(vf)
                vf = it
            outs.append((cores_of(vf, w), ne.value))
        (ra, ca), na = outs[0]
        (rb, cb_), nb = outs[1]
        same = ra == rb and all(np.array_equal(x, y) for x, y in zip(ca, cb_))
        q.put((rank, same, na, nb))
    dist.barrier()
    dist.destroy_process_group()


def _run_sharded(use_gpu, mode="vi"):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, use_gpu, mode)) for r in range(2)]
    for p in procs:
        p.start()
    import queue as _queue
    import time as _time

    res, deadline = [], _time.time() + 300
    try:
        while len(res) < 2:  # a rank that died leaves its peer in the collective: stop waiting as soon as one has failed
            try:
                res.append(q.get(timeout=1.0))
            except _queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs) or _time.time() > deadline:
                    raise AssertionError(f"sharded ranks failed or timed out: exit codes {[p.exitcode for p in procs]}")
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:  # never leave a hung peer behind
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    return sorted(res)


def test_gloo_world2_sharded_cross_driver(oracle):
    """valuef_interp_idx_sharded on two gloo ranks: every rank evaluates half of each core step's fibers, the exchange
    callback (c3sc_amd.distributed.make_fiber_exchange) all-gathers them, and both ranks end with the cores of the
    unsharded interpolation, bit for bit."""
    res = _run_sharded(False)
    for rank, same, n_sharded, n_full in res:
        assert same, f"rank {rank}: sharded result differs from the unsharded one"
        assert 0 < n_sharded < 0.75 * n_full  # about half of the fibers ran on this rank


def test_gloo_world2_sharded_policy_evaluation_fibers(oracle):
    """The same driver with bellman_pi's fibers (the greedy policy of one value function evaluated on another,
    bellman.c:1702-1886) sharded over two gloo ranks: cores identical to the unsharded interpolation on both ranks."""
    for rank, same, n_sharded, n_full in _run_sharded(False, "pi"):
        assert same, f"rank {rank}: sharded policy-evaluation result differs from the unsharded one"
        assert 0 < n_sharded < 0.75 * n_full


def test_gloo_world2_failing_rank_takes_every_rank_down_instead_of_hanging(oracle):
    """One rank's fibers fail in the middle of a sweep.  It must still enter the exchange (rows marked NaN) so that its peer,
    which is about to wait there, sees the mark: both processes end with the library's error exit -- neither hangs."""
    import time as _time

    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q, False, "fail")) for r in range(2)]
    for p in procs:
        p.start()
    deadline = _time.time() + 120
    try:
        for p in procs:
            p.join(timeout=max(1.0, deadline - _time.time()))
        codes = [p.exitcode for p in procs]
        assert all(c is not None for c in codes), f"a rank is still waiting in the exchange: exit codes {codes}"
        assert all(c == 1 for c in codes), f"both ranks must stop with the library's error exit (1): {codes}"
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)


@pytest.mark.gpu
def test_world2_sharded_policy_iteration_step_matches_single_rank(oracle):
    """c3control_step_pi on two gloo ranks sharing the box's GPU (host-driven sharded driver, consistent end points): three
    policy-evaluation sweeps end with cores bit-identical to the unsharded run on both ranks."""
    for rank, same, na, nb in _run_sharded(True, "pi"):
        assert same, f"rank {rank}: sharded c3control_step_pi differs from the unsharded one"


@pytest.mark.gpu
def test_world2_sharded_solver_step_matches_single_rank(oracle):
    """c3control_set_fiber_sharding: three value-iteration sweeps of the car problem through libc3sc.so with the fibers of
    every core step split over two ranks (both on the box's one GPU, exchange over gloo) against the same sweeps unsharded:
    identical cores on both ranks, about half the node backups per rank."""
    res = _run_sharded(True)
    for rank, same, n_sharded, n_full in res:
        assert same, f"rank {rank}: sharded sweeps differ from the unsharded ones"
        assert 0 < n_sharded < 0.8 * n_full


# ------------------------------------------------------------------------------------------------ RCCL in C (c3sc_hip_comm_*)
def _rccl_worker(rank, world, idfile, q):
    """One process per GPU: the solver sharded over the library's own RCCL communicator (c3control_shard_over_gpus), no Python in
    the exchange.  The 128-byte id travels through a file."""
    import ctypes as C
    import time

    os.environ["C3SC_HIP_DEVICE"] = str(rank if world > 1 else 0)
    import facade_lib
    from c3sc_amd.engine import load_library

    H = load_library()
    L = facade_lib.lib()
    for n in ("c3control_init_value", "c3control_step_vi"):
        getattr(L, n).restype = C.c_void_p
    L.valuef_get_ranks.restype = C.POINTER(C.c_size_t)
    L.valuef_get_cores.restype = C.POINTER(C.POINTER(C.c_double))
    idbuf = (C.c_char * 128)()
    if rank == 0:
        assert H.c3sc_hip_comm_unique_id(idbuf) == 0
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(idbuf))
        os.replace(idfile + ".tmp", idfile)
    else:
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.05)
        idbuf.raw = open(idfile, "rb").read()
    w = wl.c4_car7d().scaled(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4)
    d = w.dx

    def sweeps(sharded):
        ctl = facade_lib.Control(w, consistent_ends=None)
        if sharded:
            assert L.c3control_shard_over_gpus(ctl.h, C.c_size_t(world), C.c_size_t(rank), idbuf) == 0
        aa = C.c_void_p(L.approx_args_init())
        L.approx_args_set_cross_tol(aa, C.c_double(1e-8))
        L.approx_args_set_round_tol(aa, C.c_double(1e-8))
        L.approx_args_set_kickrank(aa, C.c_size_t(2))
        L.approx_args_set_startrank(aa, C.c_size_t(3))
        L.approx_args_set_maxrank(aa, C.c_size_t(5))
        start = facade_lib.FIBER_FN(lambda n, x, out, a: (np.ctypeslib.as_array(out, shape=(n,)).__setitem__(
            slice(None), 1.0 + 0.1 * (np.ctypeslib.as_array(x, shape=(n, d)) ** 2).sum(axis=1)), 0)[1])
        v = C.c_void_p(L.c3control_init_value(ctl.h, start, None, aa, 0))
        ne, rows = C.c_size_t(0), []
        for _ in range(3):
            nxt = C.c_void_p(L.c3control_step_vi(ctl.h, v, aa, ctl.opt, 0, C.byref(ne)))
            L.valuef_destroy(v)
            v = nxt
            ranks = [int(L.valuef_get_ranks(v)[i]) for i in range(d + 1)]
            pp = L.valuef_get_cores(v)
            rows.append((ne.value, ranks, [np.ctypeslib.as_array(pp[m], shape=(w.ngrid[m] * ranks[m] * ranks[m + 1],)).copy() for m in range(d)]))
        L.valuef_destroy(v)
        L.approx_args_free(aa)
        ctl.close()
        return rows

    full = sweeps(False)
    shard = sweeps(True)
    same = all(a[0] == b[0] and a[1] == b[1] and all(np.array_equal(x, y) for x, y in zip(a[2], b[2])) for a, b in zip(full, shard))
    # the exchange function of the host-driven driver on its own: rows [lo, hi) of a host array gathered over the communicator
    q.put((rank, bool(same), [r[0] for r in shard]))


@pytest.mark.gpu
def test_rccl_communicator_in_c_shards_the_device_resident_sweeps(tmp_path):
    """c3control_shard_over_gpus: the library's own RCCL communicator (librccl opened at run time, device buffers, stream-ordered
    all-gather inside the device-resident cross iteration) -- two ranks on two GPUs when the box has them, otherwise the same
    code path with a one-rank communicator.  Cores, ranks and node counts equal the unsharded sweeps' bit for bit on every rank."""
    import torch
    import torch.multiprocessing as mp

    world = 2 if torch.cuda.device_count() >= 2 else 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idfile = str(tmp_path / "rccl_id.bin")
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, idfile, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _queue
    import time as _time

    res, deadline = [], _time.time() + 300
    try:
        while len(res) < world:
            try:
                res.append(q.get(timeout=1.0))
            except _queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs) or _time.time() > deadline:
                    raise AssertionError(f"ranks failed or timed out: exit codes {[p.exitcode for p in procs]}")
        for p in procs:
            p.join(timeout=60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    for rank, same, counts in sorted(res):
        assert same, f"rank {rank}: sharded sweeps differ from the unsharded ones"
    print(f"RCCL communicator in C, world {world}: sharded device-resident sweeps identical to the unsharded ones; node evaluations {sorted(res)[0][2]}")


_INJECT_SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import facade_lib
from c3sc_amd import workloads as wl
from c3sc_amd.engine import load_library
H = load_library(); L = facade_lib.lib()
for n in ("c3control_init_value", "c3control_step_vi"):
    getattr(L, n).restype = C.c_void_p
w = wl.c4_car7d().scaled(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4)
ctl = facade_lib.Control(w, consistent_ends=None)
idbuf = (C.c_char * 128)()
assert H.c3sc_hip_comm_unique_id(idbuf) == 0
assert L.c3control_shard_over_gpus(ctl.h, C.c_size_t(1), C.c_size_t(0), idbuf) == 0
aa = C.c_void_p(L.approx_args_init())
L.approx_args_set_maxrank(aa, C.c_size_t(5)); L.approx_args_set_startrank(aa, C.c_size_t(3)); L.approx_args_set_kickrank(aa, C.c_size_t(2))
one = facade_lib.FIBER_FN(lambda n, x, out, a: (np.ctypeslib.as_array(out, shape=(n,)).fill(1.0), 0)[1])
v = C.c_void_p(L.c3control_init_value(ctl.h, one, None, aa, 0))
ne = C.c_size_t(0)
v = C.c_void_p(L.c3control_step_vi(ctl.h, v, aa, ctl.opt, 0, C.byref(ne)))   # clean sweep
os.environ[{var!r}] = "1"
v = C.c_void_p(L.c3control_step_vi(ctl.h, v, aa, ctl.opt, 0, C.byref(ne)))   # this rank's fibers / exchange fail: the library stops the process
print("NOT REACHED")
"""


_EXCHANGE_SCRIPT = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from c3sc_amd.engine import load_library
H = load_library()
ctx = C.c_void_p()
assert H.c3sc_hip_ctx_create(C.c_int(0), C.byref(ctx)) == 0
idbuf = (C.c_char * 128)()
assert H.c3sc_hip_comm_unique_id(idbuf) == 0
comm = C.c_void_p()
assert H.c3sc_hip_comm_create(ctx, C.c_int(1), C.c_int(0), idbuf, C.byref(comm)) == 0
H.c3sc_hip_comm_exchange.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]
F, N = 13, 7
out = np.arange(F * N, dtype=np.float64).reshape(F, N)
want = out.copy()
assert H.c3sc_hip_comm_exchange(out.ctypes.data, F, N, 0, F, comm) == 0 and np.array_equal(out, want)   # the clean exchange
os.environ["C3SC_INJECT_EXCHANGE_FAILURE"] = "1"
rc = H.c3sc_hip_comm_exchange(out.ctypes.data, F, N, 0, F, comm)
# the rank still went through the all-gather: what came back are its rows as it sent them -- marked NaN -- and the error code
assert rc == 1 and np.isnan(out).all(), (rc, out[:2])
print("ENTERED THE COLLECTIVE WITH NAN ROWS")
"""


@pytest.mark.gpu
def test_rank_failure_in_a_device_resident_sharded_sweep_enters_the_collective_and_stops():
    """A rank whose launch fails in a device-resident sharded core step (cross_device.hip: step_fibers) must still enter the RCCL
    all-gather with NaN rows and only then report the error: here with a one-rank communicator -- the failure is detected on the
    gathered array (counters[3] = 2) and the process ends with the library's error exit instead of 'NOT REACHED'."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _INJECT_SCRIPT.format(root=root, var="C3SC_INJECT_SHARD_FAILURE")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "NOT REACHED" not in r.stdout
    assert "c3sc:" in r.stderr, r.stderr[-1500:]


@pytest.mark.gpu
def test_local_failure_in_the_host_driven_exchange_still_enters_the_all_gather():
    """c3sc_hip_comm_exchange (comm_rccl.hip): a local failure before the collective (staging copy) no longer returns early -- the
    peers would wait for ever -- but sends the rank's rows as NaN through ncclAllGather and returns 1 afterwards."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _EXCHANGE_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ENTERED THE COLLECTIVE WITH NAN ROWS" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
