"""The C host side (c3sc_amd/host/libc3sc.so, headers include/c3sc/*.h) keeps the reference's API.
CPU tests: library loads, exports what the headers declare, and its host-side integer / scalar
functions agree with the pinned oracle (bit-exact for integers).  GPU tests: bellman_vi through
the reference call sequence (c3control_create ... vi_param ... bellman_vi) against the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from c3sc_amd import workloads as wl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for f in os.listdir(os.path.join(ROOT, "include", "c3sc")):
        txt = open(os.path.join(ROOT, "include", "c3sc", f)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"typedef[^;]*;", "", txt)
        names |= set(re.findall(r"\b([a-z][a-z0-9_]+)\s*\((?!\*)", txt))
    return sorted(n for n in names if n not in ("defined", "sizeof"))


def test_facade_exports_declared_symbols():
    import facade_lib

    L = facade_lib.lib()
    names = _declared()
    assert len(names) > 90
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_facade_headers_compile_as_c99():
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        open(src, "w").write('#include "c3sc/c3sc.h"\nint main(void){return 0;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", src, "-o",
                               os.path.join(td, "t.o")])


def test_keys_and_hash_match_oracle(oracle):
    import facade_lib

    L = facade_lib.lib()
    rng = np.random.default_rng(5)
    buf = C.create_string_buffer(256)
    for _ in range(200):
        n = int(rng.integers(2, 13))
        arr = rng.integers(0, 5000, size=n).astype(np.uintp)
        L.size_t_a_to_char(facade_lib.sp(arr), C.c_size_t(n), buf)
        assert buf.value == oracle.key_string(arr)
        assert L.c3sc_hashchar(1000000, buf.value) == oracle.hashchar(1000000, buf.value)
    # memo: LIFO duplicates, strcmp match (hashgrid.c:252-279)
    ht = C.c_void_p(L.htable_create(C.c_size_t(1000)))
    v1, v2 = np.array([1.5]), np.array([2.5])
    L.htable_add_element(ht, b"3 4 0 1 ", facade_lib.dp(v1), C.c_size_t(1))
    L.htable_add_element(ht, b"3 4 0 1 ", facade_lib.dp(v2), C.c_size_t(1))
    n = C.c_size_t(0)
    p = L.htable_get_element(ht, b"3 4 0 1 ", C.byref(n))
    assert n.value == 1 and p[0] == 2.5
    assert not L.htable_get_element(ht, b"3 4 0 2 ", C.byref(n))
    L.htable_destroy(ht)


@pytest.mark.parametrize("consistent", [False, True], ids=["literal", "consistent_ends"])
def test_host_integer_functions_match_oracle(oracle, consistent):
    """convert_fiber_to_ind / process_fibers_neighbor of the C host side vs the oracle: bit-exact -- with the reference's literal
    end-point rule and with the solver's consistent one (boundary_set_consistent_ends / orc_boundary_set_consistent_ends)."""
    import facade_lib

    L = facade_lib.lib()
    w = wl.c4_car7d().scaled(ngrid=(6, 5, 7, 4, 5, 6, 5), rank=3)
    P = oracle.Problem(w)
    ctl = facade_lib.Control(w)
    xg = ctl.xgrid()
    for m in range(w.dx):
        np.testing.assert_array_equal(xg[m], P.xgrid(m))
    bnd = oracle.Boundary(w.lb, w.ub)
    fb = C.c_void_p(L.boundary_alloc(C.c_size_t(w.dx), facade_lib.dp(facade_lib.f64(w.lb)), facade_lib.dp(facade_lib.f64(w.ub))))
    for m, nme in enumerate(w.bc_names()):
        bnd.set_type(m, nme)
        L.boundary_external_set_type(fb, C.c_size_t(m), nme.encode())
    for cen, wid in w.obstacles:
        bnd.add_obstacle(cen, wid)
        L.boundary_add_obstacle(fb, facade_lib.dp(facade_lib.f64(cen)), facade_lib.dp(facade_lib.f64(wid)))
    if consistent:
        oracle.lib().orc_boundary_set_consistent_ends(bnd.h, C.c_int(1))
        L.boundary_set_consistent_ends(fb, C.c_int(1))
    ng = facade_lib.usz(w.ngrid)
    kept = 0
    for k in range(w.dx):
        idx = wl.synth_fibers(w, k, 40)
        idx[0, :] = 0
        idx[1, :] = np.array(w.ngrid) - 1
        for row in idx:
            N = w.ngrid[k]
            x = np.array([[xg[m][j] if m == k else xg[m][row[m]] for m in range(w.dx)] for j in range(N)])
            r0, fi0, dv0 = oracle.convert_fiber_to_ind(x, w.ngrid, xg)
            fi = np.zeros(w.dx, dtype=np.uintp)
            dv = C.c_size_t(0)
            r1 = L.convert_fiber_to_ind(C.c_size_t(w.dx), C.c_size_t(N), facade_lib.dp(x), facade_lib.sp(ng), facade_lib.ptrs(xg),
                                        facade_lib.sp(fi), C.byref(dv))
            assert (r0, dv0) == (r1, dv.value) and (fi0 == fi).all()
            _, ab0, nv0, nf0 = oracle.process_fibers_neighbor(fi0, k, x, w.ngrid, bnd)
            ab = np.zeros(N, dtype=np.int32)
            nv = np.zeros(2 * N, dtype=np.uintp)
            nf = np.zeros(2 * (w.dx - 1), dtype=np.uintp)
            L.process_fibers_neighbor(C.c_size_t(w.dx), facade_lib.sp(fi), C.c_size_t(k), facade_lib.dp(x),
                                      ab.ctypes.data_as(C.POINTER(C.c_int)), facade_lib.sp(nv), facade_lib.sp(nf), facade_lib.sp(ng), fb)
            np.testing.assert_array_equal(ab, ab0)
            np.testing.assert_array_equal(nv, nv0)
            np.testing.assert_array_equal(nf, nf0)
            kept += int(w.bc[k] != wl.BC_ABSORB and (ab[0] != 0 or ab[-1] != 0))
    assert (kept > 0) == consistent  # the literal rule never leaves a flag on a reflecting / periodic fiber's end point
    L.boundary_free(fb)
    ctl.close()


def test_host_transition_and_rhs_match_oracle(oracle):
    import facade_lib

    L = facade_lib.lib()
    rng = np.random.default_rng(9)
    for dx, du in ((2, 1), (3, 3), (7, 2)):
        for _ in range(50):
            tv = rng.uniform(1e-3, 1.0, 2 * dx)
            drift = rng.uniform(-2, 2, dx)
            drift[rng.integers(0, dx)] = 0.0  # dead zone
            diff = np.diag(rng.uniform(0.1, 1.0, dx)).ravel()
            gd = rng.uniform(-1, 1, dx * du)
            gdiff = rng.uniform(-0.1, 0.1, dx * dx * du)
            h2 = 1e-3
            res0, p0, dt0, gp0, gdt0 = oracle.transition_assemble(dx, du, dx, h2, tv, drift, diff, gd, gdiff)
            p = np.zeros(2 * dx + 1); gp = np.zeros((2 * dx + 1) * du); gdt = np.zeros(du); sp_ = np.zeros(du)
            dt = C.c_double(0)
            res = L.transition_assemble(dx, du, dx, h2, facade_lib.dp(tv), facade_lib.dp(drift), facade_lib.dp(gd), facade_lib.dp(diff),
                                        facade_lib.dp(gdiff), facade_lib.dp(p), facade_lib.dp(gp), C.cast(C.byref(dt), facade_lib.c_double_p),
                                        facade_lib.dp(gdt), facade_lib.dp(sp_))
            assert res == res0
            np.testing.assert_allclose(p, p0, rtol=0, atol=1e-16)
            np.testing.assert_allclose(gp, gp0, rtol=1e-14, atol=1e-15)
            assert dt.value == pytest.approx(dt0, rel=1e-15)
            cost = rng.uniform(0, 5, 2 * dx + 1)
            sg = rng.uniform(-1, 1, du)
            g = np.zeros(du)
            v = L.bellmanrhs(dx, du, 0.7, facade_lib.dp(sg), 0.1, facade_lib.dp(p), facade_lib.dp(gp), dt.value, facade_lib.dp(gdt),
                             facade_lib.dp(cost), facade_lib.dp(g))
            v0, g0 = oracle.bellmanrhs(dx, du, 0.7, 0.1, p0, dt0, cost, sg, gp0, gdt0)
            assert v == pytest.approx(v0, rel=1e-15)
            np.testing.assert_allclose(g, g0, rtol=1e-13, atol=1e-15)
    # util.c:995-1006: largest s with s*(M-1) < N-1, minus nothing more -- N == M yields 0 (reference behaviour)
    assert L.uniform_stride(C.c_size_t(51), C.c_size_t(5)) == 12 and L.uniform_stride(C.c_size_t(10), C.c_size_t(10)) == 0


def test_old_assembly_dyn_pair_and_periodic_wrap(oracle):
    """transition_assemble_old (nodeutil.c:82-233) against the oracle, including the doubled left update of its
    zero-drift gradient branch; dyn_* (dynamics.c:258-354); outer_bound_dim (boundary.c:577-597); diag_create."""
    import facade_lib

    L = facade_lib.lib()
    rng = np.random.default_rng(21)
    L.transition_assemble_old.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_double] + [facade_lib.c_double_p] * 10
    for dx, du in ((2, 1), (3, 2), (7, 3)):
        for rep in range(40):
            hv = rng.uniform(0.02, 0.3, dx)
            hmin = float(hv.min())
            drift = rng.uniform(-2, 2, dx)
            drift[rng.integers(0, dx)] = 0.0
            if rep % 5 == 0:
                drift[:] = 0.0
            diff = np.diag(rng.uniform(0.1, 1.0, dx)).ravel()
            gd = rng.uniform(-1, 1, dx * du)
            gd[rng.integers(0, dx * du)] = 0.0
            gdiff = rng.uniform(-0.1, 0.1, dx * dx * du)
            for grad in (True, False):
                res0, p0, dt0, gp0, gdt0 = oracle.transition_assemble(dx, du, dx, hmin, hv, drift, diff, gd if grad else None,
                                                                      gdiff if grad else None, old=True)
                p = np.zeros(2 * dx + 1); gp = np.zeros((2 * dx + 1) * du); gdt = np.zeros(du); sp_ = np.zeros(du)
                dt = C.c_double(0)
                res = L.transition_assemble_old(dx, du, dx, hmin, facade_lib.dp(hv), facade_lib.dp(drift), facade_lib.dp(gd) if grad else None,
                                                facade_lib.dp(diff), facade_lib.dp(gdiff) if grad else None, facade_lib.dp(p),
                                                facade_lib.dp(gp) if grad else None, C.cast(C.byref(dt), facade_lib.c_double_p),
                                                facade_lib.dp(gdt) if grad else None, facade_lib.dp(sp_) if grad else None)
                assert res == res0
                np.testing.assert_allclose(p, p0, rtol=0, atol=1e-16)
                assert dt.value == pytest.approx(dt0, rel=1e-15)
                if grad:
                    np.testing.assert_allclose(gp, gp0, rtol=1e-14, atol=1e-15)
                    np.testing.assert_allclose(gdt, gdt0, rtol=1e-14, atol=1e-18)
    # stationary node: return 1 and leave the outputs alone (nodeutil.c:199-201)
    p = np.full(5, 7.0)
    dt = C.c_double(3.0)
    z2 = np.zeros(2)
    res = L.transition_assemble_old(2, 1, 2, 0.1, facade_lib.dp(np.array([0.1, 0.2])), facade_lib.dp(z2), None, facade_lib.dp(np.zeros(4)), None,
                                    facade_lib.dp(p), None, C.cast(C.byref(dt), facade_lib.c_double_p), None, None)
    assert res == 1 and dt.value == 3.0 and p[4] == 7.0

    # drift + diffusion pair
    for n in ("drift_copy", "diff_copy", "dyn_alloc", "dyn_copy_deep", "diag_create"):
        getattr(L, n).restype = C.c_void_p
    for n in ("dyn_get_dx", "dyn_get_du", "dyn_get_dw", "diag_count"):
        getattr(L, n).restype = C.c_size_t
    calls = []

    def b(t, x, u, out, jac, args):
        calls.append("b")
        out[0], out[1] = x[1], u[0]
        if jac:
            jac[0], jac[1] = 0.0, 1.0
        return 0

    def bfail(t, x, u, out, jac, args):
        calls.append("bfail")
        return 5

    def s(t, x, u, out, jac, args):
        calls.append("s")
        out[0], out[1], out[2], out[3] = 0.5, 0.0, 0.0, 0.25
        return 0

    bcb, bfcb, scb = facade_lib.DYN_FN(b), facade_lib.DYN_FN(bfail), facade_lib.DYN_FN(s)
    dr = C.c_void_p(L.drift_alloc(C.c_size_t(2), C.c_size_t(1)))
    df = C.c_void_p(L.diff_alloc(C.c_size_t(2), C.c_size_t(1), C.c_size_t(2)))
    L.drift_add_func(dr, bcb, None)
    L.diff_add_func(df, scb, None)
    dyn = C.c_void_p(L.dyn_alloc(dr, df))
    assert (L.dyn_get_dx(dyn), L.dyn_get_du(dyn), L.dyn_get_dw(dyn)) == (2, 1, 2)
    deep = C.c_void_p(L.dyn_copy_deep(dyn))
    x, u = np.array([0.3, -0.7]), np.array([0.9])
    for h in (dyn, deep):
        bo, jo, so = np.zeros(2), np.zeros(2), np.zeros(4)
        assert L.dyn_eval(h, C.c_double(0.0), facade_lib.dp(x), facade_lib.dp(u), facade_lib.dp(bo), facade_lib.dp(jo), facade_lib.dp(so), None) == 0
        assert list(bo) == [-0.7, 0.9] and list(jo) == [0.0, 1.0] and list(so) == [0.5, 0.0, 0.0, 0.25]
    calls.clear()
    so = np.zeros(4)
    assert L.dyn_eval(dyn, C.c_double(0.0), facade_lib.dp(x), facade_lib.dp(u), None, None, facade_lib.dp(so), None) == 0 and calls == ["s"]
    L.drift_add_func(dr, bfcb, None)  # a failing drift stops before the diffusion (dynamics.c:341-343)
    calls.clear()
    bo = np.zeros(2)
    assert L.dyn_eval(dyn, C.c_double(0.0), facade_lib.dp(x), facade_lib.dp(u), facade_lib.dp(bo), None, facade_lib.dp(so), None) == 5
    assert calls == ["bfail"]
    L.dyn_free_deep(deep)
    L.dyn_free(dyn)
    L.drift_free(dr)
    L.diff_free(df)

    # periodic wrap
    L.outer_bound_dim.restype = C.c_double
    lb, ub = np.array([-1.0, 0.0]), np.array([2.0, 6.0])
    bd = C.c_void_p(L.boundary_alloc(C.c_size_t(2), facade_lib.dp(lb), facade_lib.dp(ub)))
    L.boundary_external_set_type(bd, C.c_size_t(1), b"periodic")
    mp = C.c_int(-1)
    for dim, xx, want, wmap in ((1, 0.0, 6.0, 1), (1, -0.5, 6.0, 1), (1, 6.0, 0.0, 2), (1, 7.5, 0.0, 2), (1, 3.0, 3.0, 0),
                                (0, -1.0, -1.0, 0), (0, 2.5, 2.5, 0), (0, 0.0, 0.0, 0)):
        got = L.outer_bound_dim(bd, C.c_size_t(dim), C.c_double(xx), C.byref(mp))
        assert (got, mp.value) == (want, wmap)
    L.boundary_free(bd)

    # one unlinked Diag record
    rk = np.array([1, 4, 5, 1], dtype=np.uintp)
    dg = C.c_void_p(L.diag_create(C.c_size_t(3), C.c_int(1), C.c_double(2.0), C.c_double(0.5), C.c_size_t(3), facade_lib.sp(rk), C.c_double(0.1)))
    assert L.diag_count(dg) == 1
    L.diag_append(C.byref(dg), C.c_size_t(4), C.c_int(1), C.c_double(2.0), C.c_double(0.25), C.c_size_t(3), facade_lib.sp(rk), C.c_double(0.1))
    assert L.diag_count(dg) == 2
    L.diag_destroy(C.byref(dg))
    assert dg.value is None


def test_integer_memo_fiber_keys_agree_with_node_keys():
    """The integer-keyed twin of the node memo: a node stored through a fiber along one dimension is found through a
    fiber along another and through the plain per-node key (the property the string keys of bellman.c:1333-1344 have);
    growth keeps every entry; clearing forgets them."""
    import facade_lib

    L = facade_lib.lib()
    L.fastmemo_create.restype = C.c_void_p
    L.fastmemo_size.restype = C.c_size_t

    class FmFiber(C.Structure):
        _fields_ = [("key", C.c_uint64 * 4), ("base", C.c_uint64), ("pre0", C.c_uint64), ("step", C.c_uint64), ("word", C.c_uint), ("shift", C.c_uint)]

    m = C.c_void_p(L.fastmemo_create())
    rng = np.random.default_rng(5)
    d, N = 7, 41
    ff, key, val = FmFiber(), (C.c_uint64 * 4)(), C.c_double(0)
    truth = {}
    for _ in range(4000):  # 4000 x 41 nodes: several doublings of the initial 65536 slots
        idx = rng.integers(0, N, d).astype(np.int32)
        k = int(rng.integers(0, d))
        L.fastmemo_fiber_begin(C.byref(ff), C.c_size_t(d), idx.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(k), C.c_uint64(0), C.c_uint64(3))
        for j in range(N):
            node = tuple(int(v) for v in idx[:k]) + (j,) + tuple(int(v) for v in idx[k + 1:])
            v = float(rng.standard_normal())
            L.fastmemo_fiber_put(m, C.byref(ff), C.c_size_t(j), C.c_double(v))
            truth.setdefault(node, v)  # a repeated key keeps its first value
    assert L.fastmemo_size(m) == len(truth)
    nodes = list(truth)
    for t in rng.permutation(len(nodes))[:3000]:
        node = nodes[t]
        idx = np.array(node, dtype=np.int32)
        k = int(rng.integers(0, d))  # look it up along any dimension
        L.fastmemo_fiber_begin(C.byref(ff), C.c_size_t(d), idx.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(k), C.c_uint64(0), C.c_uint64(3))
        assert L.fastmemo_fiber_get(m, C.byref(ff), C.c_size_t(node[k]), C.byref(val)) == 1 and val.value == truth[node]
        L.fastmemo_key(C.c_size_t(d), idx.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(k), C.c_size_t(node[k]), C.c_uint64(0), C.c_uint64(3), key)
        assert L.fastmemo_get(m, key, C.byref(val)) == 1 and val.value == truth[node]
        # another counter is another key
        L.fastmemo_fiber_counter(C.byref(ff), C.c_uint64(0), C.c_uint64(4))
        assert L.fastmemo_fiber_get(m, C.byref(ff), C.c_size_t(node[k]), C.byref(val)) == 0
    L.fastmemo_clear(m)
    assert L.fastmemo_size(m) == 0
    idx = np.array(nodes[0], dtype=np.int32)
    L.fastmemo_key(C.c_size_t(d), idx.ctypes.data_as(C.POINTER(C.c_int32)), C.c_size_t(0), C.c_size_t(nodes[0][0]), C.c_uint64(0), C.c_uint64(3), key)
    assert L.fastmemo_get(m, key, C.byref(val)) == 0
    L.fastmemo_put(m, key, C.c_double(2.5))
    assert L.fastmemo_get(m, key, C.byref(val)) == 1 and val.value == 2.5 and L.fastmemo_size(m) == 1
    L.fastmemo_free(m)


def test_boundinfo_hashgrid_and_small_helpers():
    """The remaining public helpers of boundary.h / util.h: BoundInfo queries (boundary.c:491-801), HashGrid
    (util.c:352-657), c3sc_check_bounds / combine_and_sort / sample_discrete_rv (util.c:225-331)."""
    import facade_lib

    L = facade_lib.lib()
    for n in ("boundary_type", "hash_grid_create", "hash_grid_create_grid", "hash_grid_create_ndgrid", "bound_info_alloc"):
        getattr(L, n).restype = C.c_void_p
    L.bound_info_period_xmap.restype = C.c_double
    L.hash_grid_get_ind.restype = C.c_size_t
    L.c3sc_sample_discrete_rv.restype = C.c_size_t
    L.c3sc_combine_and_sort.restype = facade_lib.c_double_p

    # ---- BoundInfo
    lb, ub = np.array([-1.0, 0.0, -2.0]), np.array([1.0, 6.0, 2.0])
    bd = C.c_void_p(L.boundary_alloc(C.c_size_t(3), facade_lib.dp(lb), facade_lib.dp(ub)))  # all absorbing
    L.boundary_external_set_type(bd, C.c_size_t(1), b"periodic")
    L.boundary_external_set_type(bd, C.c_size_t(2), b"reflect")
    L.boundary_add_obstacle(bd, facade_lib.dp(np.array([0.0, 3.0, 0.0])), facade_lib.dp(np.array([0.5, 0.5, 0.5])))

    def info(x):
        bi = C.c_void_p(L.boundary_type(bd, C.c_double(0.0), facade_lib.dp(f64(x))))
        r = dict(on=L.bound_info_onbound(bi), absorb=L.bound_info_absorb(bi), period=L.bound_info_period(bi),
                 reflect=L.bound_info_reflect(bi), obs=L.bound_info_get_in_obstacle(bi),
                 ondim=[L.bound_info_onbound_dim(bi, C.c_size_t(m)) for m in range(3)],
                 pdir=L.bound_info_period_dim_dir(bi, C.c_size_t(1)), rdir=L.bound_info_reflect_dim_dir(bi, C.c_size_t(2)),
                 xmap=L.bound_info_period_xmap(bi, C.c_size_t(1)))
        L.bound_info_free(bi)
        return r

    f64 = facade_lib.f64
    r = info([0.5, 1.0, 1.0])
    assert (r["on"], r["absorb"], r["period"], r["reflect"], r["obs"], r["ondim"]) == (0, 0, 0, 0, -1, [0, 0, 0])
    r = info([-1.0, 1.0, 1.0])  # on the absorbing left face of dim 0 (faces are closed)
    assert (r["on"], r["absorb"], r["ondim"]) == (1, 1, [1, 0, 0])
    r = info([0.5, 6.5, 1.0])  # past the periodic right face: image is the left bound
    assert (r["on"], r["absorb"], r["period"], r["pdir"], r["xmap"]) == (1, 0, 1, 1, 0.0)
    r = info([0.5, 0.0, 1.0])
    assert (r["period"], r["pdir"], r["xmap"]) == (1, -1, 6.0)
    r = info([0.5, 1.0, -2.0])
    assert (r["on"], r["absorb"], r["reflect"], r["rdir"], r["ondim"]) == (1, 0, 1, -1, [0, 0, 1])
    r = info([0.1, 3.2, -0.2])  # inside the obstacle: absorbed, every dimension reports "on boundary"
    assert (r["on"], r["absorb"], r["obs"], r["ondim"]) == (1, 1, 0, [1, 1, 1])
    bi = C.c_void_p(L.bound_info_alloc(C.c_size_t(2)))
    assert L.bound_info_set_dim(bi, C.c_int(1), C.c_int(2), C.c_size_t(0)) == 1  # LEFT, PERIODIC: image still owed
    assert L.bound_info_set_dim(bi, C.c_int(2), C.c_int(3), C.c_size_t(1)) == 0  # RIGHT, REFLECT
    assert L.bound_info_set_dim(bi, C.c_int(2), C.c_int(7), C.c_size_t(1)) == -1
    L.bound_info_free(bi)
    L.boundary_free(bd)

    # ---- HashGrid
    class Vec(C.Structure):
        _fields_ = [("size", C.c_size_t), ("elem", facade_lib.c_double_p)]

    g0, g1 = np.linspace(-1.0, 1.0, 41), np.linspace(0.0, 2 * np.pi, 101)
    v0, v1 = Vec(41, facade_lib.dp(g0)), Vec(101, facade_lib.dp(g1))
    hg = C.c_void_p(L.hash_grid_create_grid(C.c_size_t(17), C.byref(v0)))  # fewer buckets than values: chains
    ex = C.c_int(-1)
    for i in (0, 7, 20, 40):
        assert L.hash_grid_get_ind(hg, C.c_double(g0[i]), C.byref(ex)) == i and ex.value == 1
    assert L.hash_grid_get_ind(hg, C.c_double(np.nextafter(g0[7], 1.0)), C.byref(ex)) == 0 and ex.value == 0  # exact values only
    assert L.hash_grid_get_ind(hg, C.c_double(-0.0), C.byref(ex)) == 20 and ex.value == 1
    assert L.hash_grid_add_element(hg, C.c_size_t(99), C.c_double(g0[3])) == 2
    assert L.hash_grid_add_element(hg, C.c_size_t(41), C.c_double(5.0)) == 0
    assert L.hash_grid_get_ind(hg, C.c_double(5.0), C.byref(ex)) == 41
    L.hash_grid_free(hg)
    vecs = (C.POINTER(Vec) * 2)(C.pointer(v0), C.pointer(v1))
    nd = C.c_void_p(L.hash_grid_create_ndgrid(C.c_size_t(1000), C.c_size_t(2), vecs))
    out = np.zeros(2, dtype=np.uintp)
    assert L.hash_grid_ndgrid_get_ind(nd, C.c_size_t(2), facade_lib.dp(np.array([g0[5], g1[77]])), facade_lib.sp(out)) == 0
    assert list(out) == [5, 77]
    assert L.hash_grid_ndgrid_get_ind(nd, C.c_size_t(2), facade_lib.dp(np.array([g0[5], 0.123])), facade_lib.sp(out)) == 1
    L.hash_grid_free_ndgrid(C.c_size_t(2), nd)
    assert L.hash_grid_create(C.c_size_t(0)) is None

    # ---- small helpers
    lo, hi = np.array([0.0, 0.0, 0.0]), np.array([1.0, 2.0, 3.0])
    chk = lambda x: L.c3sc_check_bounds(C.c_size_t(3), facade_lib.dp(lo), facade_lib.dp(hi), facade_lib.dp(f64(x)))
    assert (chk([0.5, 1, 1]), chk([0.5, -1e-9, 5]), chk([0.5, 2, 3.5]), chk([0, 2, 3])) == (0, -2, 3, 0)
    assert L.c3sc_check_bounds(C.c_size_t(3), None, facade_lib.dp(hi), facade_lib.dp(f64([9, 9, 9]))) == 0
    x, y, nt = np.array([3.0, 1.0, 2.0]), np.array([2.0, 0.5, 3.0 + 1e-16, 4.0]), C.c_size_t(0)
    p = L.c3sc_combine_and_sort(C.c_size_t(3), facade_lib.dp(x), C.c_size_t(4), facade_lib.dp(y), C.byref(nt))
    assert nt.value == 5 and list(np.ctypeslib.as_array(p, shape=(5,))) == [0.5, 1.0, 2.0, 3.0, 4.0]
    C.CDLL(None).free(p)
    pr = np.array([0.5, 0.1, 0.4])  # ascending: 0.1 (ind 1), 0.4 (ind 2), 0.5 (ind 0) -> sums 0.1, 0.5, 1.0
    picks = []
    for u in (0.05, 0.1, 0.3, 0.5, 0.75, 1.0):
        q = pr.copy()
        picks.append(L.c3sc_sample_discrete_rv(C.c_size_t(3), facade_lib.dp(q), C.c_double(u)))
    assert picks == [1, 1, 2, 2, 0, 0] and np.allclose(q, [0.1, 0.5, 1.0])

    # ---- cross index handle / workspace scratch lists exist
    L.valuef_get_isl.restype = C.c_void_p
    L.workspace_get_absorbed_no.restype = C.c_void_p
    L.workspace_get_absorbed_yes.restype = C.c_void_p
    w = C.c_void_p(L.workspace_alloc(C.c_size_t(2), C.c_size_t(1), C.c_size_t(2), C.c_size_t(11)))
    assert L.workspace_get_absorbed_no(w) and L.workspace_get_absorbed_yes(w)
    L.workspace_free(w)


def _callbacks(w):
    """Host callbacks with the reference's signatures, evaluated in Python (dubins: dubinscar.c:40-121)."""
    import math

    import facade_lib

    def drift(t, x, u, out, jac, args):
        out[0], out[1], out[2] = math.cos(x[2]), math.sin(x[2]), u[0]
        return 0

    def diff(t, x, u, out, grad, args):
        for i in range(9):
            out[i] = 0.0
        out[0], out[4], out[8] = 1.0, 1.0, 1e-2
        return 0

    def stage(t, x, u, out, grad):
        out[0] = 1.0
        return 0

    def bcost(t, x, out):
        out[0] = 10.0
        return 0

    def ocost(x, out):
        out[0] = 0.0
        return 0

    return (facade_lib.DYN_FN(drift), facade_lib.DYN_FN(diff), facade_lib.STAGE_FN(stage), facade_lib.BOUND_FN(bcost),
            facade_lib.OBS_FN(ocost))


@pytest.mark.gpu
def test_bellman_vi_through_reference_api(oracle):
    """c3control_create -> add_* -> begin_vi (control_params/vi_param/vi_iter) -> bellman_vi(N, x, out, vi):
    values vs the oracle's bellman_vi, memo hits on the second call, device model cross-checked
    against the host callbacks."""
    import facade_lib

    w = wl.c2_dubins().scaled(ngrid=(21, 17, 16), rank=4)
    cores = wl.synth_cores(w)
    P = oracle.Problem(w, cores)
    P.increment_vi_iter()
    ctl = facade_lib.Control(w, _callbacks(w))
    vf = ctl.valuef(cores)
    vi = ctl.begin_vi(vf)
    xg = ctl.xgrid()
    total = 0
    for k in range(3):
        idx = wl.synth_fibers(w, k, 12)
        idx[0, :] = 0
        for row in idx:
            N = w.ngrid[k]
            x = np.array([[xg[m][j] if m == k else xg[m][row[m]] for m in range(3)] for j in range(N)])
            ref, _ = P.bellman_vi(x, use_memo=True)
            out = ctl.bellman_vi(vi, x)
            assert np.abs(out - ref).max() <= 1e-12 * np.abs(ref).max()
            total += N
    # memo: same counts as the reference's nnode_evals; a repeated fiber is served from the table
    assert ctl.nnode_evals(vi) == P.nnode_evals()
    before = ctl.nnode_evals(vi)
    out2 = ctl.bellman_vi(vi, x)
    np.testing.assert_array_equal(out2, out)
    assert ctl.nnode_evals(vi) == before
    assert ctl.end_vi(vi) == before
    ctl.close()


@pytest.mark.gpu
def test_bellman_vi_unchanged_example_callbacks(oracle):
    """No device model registered: bellman_vi evaluates the user's host callbacks (as the reference's examples
    define them) into tables and still runs the backup on the GPU -- the 'examples link unchanged' path."""
    import facade_lib

    w = wl.c2_dubins().scaled(ngrid=(15, 14, 16), rank=6)
    cores = wl.synth_cores(w)
    P = oracle.Problem(w, cores)
    P.increment_vi_iter()  # c3control_begin_vi does the same (bellman.c:2196); the memo key carries the iteration
    ctl = facade_lib.Control(w, _callbacks(w), device_model=False)
    vf = ctl.valuef(cores)
    vi = ctl.begin_vi(vf)
    xg = ctl.xgrid()
    for k in range(3):
        idx = wl.synth_fibers(w, k, 20)
        idx[0, :] = 0
        idx[1, :] = np.array(w.ngrid) - 1
        idx[:, k] = 0
        N = w.ngrid[k]
        x = np.array([[[xg[m][j] if m == k else xg[m][row[m]] for m in range(3)] for j in range(N)] for row in idx])
        out = ctl.bellman_vi_batch(vi, x)
        # reference semantics incl. the per-iteration memo: a node already served along another fiber keeps
        # its first value even where the end-point quirk (nodeutil.c:570-612) would now flag it differently
        ref = np.array([P.bellman_vi(xf, use_memo=True)[0] for xf in x])
        assert np.abs(out - ref).max() <= 1e-12 * np.abs(ref).max()
    ctl.end_vi(vi)
    ctl.close()


@pytest.mark.gpu
def test_bellman_vi_batch_and_fiber_nn(oracle):
    """One launch for many callback fibers (bellman_vi_batch) and the literal valuef_eval_fiber_ind_nn
    interface (explicit neighbour arrays, tprob_test.c:575-602 style) on the GPU."""
    import facade_lib

    L = facade_lib.lib()
    w = wl.c4_car7d().scaled(ngrid=(9, 8, 10, 7, 6, 5, 11), rank=4)
    cores = wl.synth_cores(w)
    P = oracle.Problem(w, cores)
    ctl = facade_lib.Control(w)
    vf = ctl.valuef(cores)
    vi = ctl.begin_vi(vf)
    xg = ctl.xgrid()
    k = 2
    idx = wl.synth_fibers(w, k, 50)
    N = w.ngrid[k]
    x = np.array([[[xg[m][j] if m == k else xg[m][row[m]] for m in range(w.dx)] for j in range(N)] for row in idx])
    out = ctl.bellman_vi_batch(vi, x)
    ref, _, _ = P.bellman_fibers(k, idx)
    assert np.abs(out - ref).max() <= 1e-12 * np.abs(ref).max()
    # valuef_eval_fiber_ind_nn with caller-chosen neighbours
    fixed = np.array([3, 4, 0, 2, 1, 3, 5], dtype=np.uintp)
    nbf = np.array([1, 4, 4, 6, 0, 3, 5, 5, 0, 2, 9, 10], dtype=np.uintp)
    nbv = np.zeros(2 * N, dtype=np.uintp)
    for j in range(N):
        nbv[2 * j], nbv[2 * j + 1] = (j + 3) % N, (j * 7) % N
    got = np.zeros(N * (2 * w.dx + 1))
    rc = L.valuef_eval_fiber_ind_nn(vf, facade_lib.sp(fixed), C.c_size_t(k), facade_lib.sp(nbf), facade_lib.sp(nbv), facade_lib.dp(got))
    assert rc == 0
    want = P.vf.eval_fiber_ind_nn(fixed, k, nbf, nbv).ravel()
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    ctl.end_vi(vi)
    ctl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("device_model", [True, False], ids=["device_model", "host_callbacks"])
def test_bellman_pi_through_reference_api(oracle, device_model):
    """c3control_begin_pi (pi_solve head) -> begin_pi_step (step_pi state) -> bellman_pi(N, x, out, poli) with the
    policy greedy for one value function and the iterate another: values and the reference's counters
    (npol_evals: policy computed once per node and pi_iter; niter_node_evals: every node of every call, Q2)."""
    import facade_lib

    w = wl.c2_dubins().scaled(ngrid=(21, 17, 16), rank=4)
    cores_pol = wl.synth_cores(w)
    cores_it = [c * (1.0 + 0.05 * np.cos(np.arange(c.size)).reshape(c.shape)) for c in wl.smooth_cores(w)]
    P = oracle.Problem(w, cores_it)
    pol_vf = oracle.ValueF(w.ngrid, w.ranks, cores_pol)
    ctl = facade_lib.Control(w, _callbacks(w), device_model=device_model)
    vf_pol, vf_it = ctl.valuef(cores_pol), ctl.valuef(cores_it)
    pi = ctl.begin_pi(vf_pol)
    P.pi_begin()
    xg = ctl.xgrid()
    L = facade_lib.lib()
    for step in range(2):  # second step: new pi_subiter, same policy -> no new policy evaluations
        ctl.begin_pi_step(pi, vf_it)
        P.pi_step_begin()
        for k in range(3):
            idx = wl.synth_fibers(w, k, 10)
            idx[0, :] = 0
            idx[1, :] = np.array(w.ngrid) - 1
            N = w.ngrid[k]
            x = np.array([[[xg[m][j] if m == k else xg[m][row[m]] for m in range(3)] for j in range(N)] for row in idx])
            ref = np.array([P.bellman_pi(pol_vf, xf)[0] for xf in x])
            if k == 1:
                out = np.array([ctl.bellman_pi(pi, xf) for xf in x])  # the callback ABI, one fiber per call
            else:
                out = ctl.bellman_pi_batch(pi, x)
            assert np.abs(out - ref).max() <= 1e-12 * np.abs(ref).max()
        assert L.pi_param_get_npol_evals(pi) == P.npol_evals()
        assert ctl.end_pi_step(pi) == P.niter_node_evals()
    L.pi_param_destroy(pi)
    ctl.close()


def _lqg2d_callbacks():
    """tprob_test.c f1b / s1 / stagecost2d / boundcost / ocost (lines 132-318)."""
    import facade_lib

    def drift(t, x, u, out, jac, args):
        out[0], out[1] = x[1], u[0]
        return 0

    def diff(t, x, u, out, grad, args):
        out[0], out[1], out[2], out[3] = 1.0, 0.0, 0.0, 1.0
        return 0

    def stage(t, x, u, out, grad):
        out[0] = x[0] * x[0] + x[1] * x[1] + u[0] * u[0]
        return 0

    def bcost(t, x, out):
        out[0] = 100.0
        return 0

    def ocost(x, out):
        out[0] = 0.0
        return 0

    return (facade_lib.DYN_FN(drift), facade_lib.DYN_FN(diff), facade_lib.STAGE_FN(stage), facade_lib.BOUND_FN(bcost),
            facade_lib.OBS_FN(ocost))


@pytest.mark.gpu
def test_continuous_controls_through_reference_api(oracle):
    """c3opt_alloc(BFGS) + add_lb/ub as the reference's tests and examples set it up (tprob_test.c:2291-2299):
    bellman_vi minimises over the control box on the device; the device result is cross-checked against the host
    box minimiser over the user's callbacks on the first fiber (cross_check_model), obeys the reference's own
    property (not above the 100-point grid minimum + 1e-10, tprob_test.c:1540), and value/policy iteration run."""
    import facade_lib

    L = facade_lib.lib()
    for n in ("c3control_init_value", "c3control_vi_solve", "c3control_pi_solve"):
        getattr(L, n).restype = C.c_void_p
    L.valuef_norm2diff.restype = C.c_double
    w0 = wl.c1_lqg2d().scaled(ngrid=(25, 23), rank=4)
    cores = wl.synth_cores(w0)
    ctl = facade_lib.Control(w0, _lqg2d_callbacks(), box=([-1.0], [1.0]))
    vf = ctl.valuef(cores)
    vi = ctl.begin_vi(vf)
    xg = ctl.xgrid()
    w100 = wl.Workload(w0.name, w0.model, w0.params, w0.dx, w0.du, w0.lb, w0.ub, w0.ngrid, w0.ranks, w0.discount, w0.bc, [],
                       np.linspace(-1, 1, 100).reshape(-1, 1))
    P100 = oracle.Problem(w100, cores)
    for k in range(2):
        idx = wl.synth_fibers(w0, k, 12)
        N = w0.ngrid[k]
        x = np.array([[[xg[m][j] if m == k else xg[m][row[m]] for m in range(2)] for j in range(N)] for row in idx])
        out = ctl.bellman_vi_batch(vi, x)
        ref, _, ab = P100.bellman_fibers(k, idx)
        assert (out[ab == 0] <= ref[ab == 0] + 1e-10).all()
        assert (out[ab == 0] >= ref[ab == 0] - 1e-3 * np.abs(ref).max()).all()
        np.testing.assert_allclose(out[ab != 0], ref[ab != 0], rtol=1e-12)
    ctl.end_vi(vi)
    ctl.close()
    # the solver loops with continuous controls (stronger discount so that a few updates visibly contract)
    w = wl.Workload(w0.name, w0.model, w0.params, w0.dx, w0.du, w0.lb, w0.ub, w0.ngrid, w0.ranks, 8.0, w0.bc, [], w0.cands)
    ctl = facade_lib.Control(w, box=([-1.0], [1.0]))
    aa = C.c_void_p(L.approx_args_init())
    L.approx_args_set_maxrank(aa, C.c_size_t(12))
    L.approx_args_set_startrank(aa, C.c_size_t(3))
    L.approx_args_set_kickrank(aa, C.c_size_t(3))
    const = facade_lib.FIBER_FN(lambda n, x, out, a: (np.ctypeslib.as_array(out, shape=(n,)).fill(0.2), 0)[1])
    cost = C.c_void_p(L.c3control_init_value(ctl.h, const, None, aa, 0))
    diffs = []
    for upd in range(4):
        nxt = C.c_void_p(L.c3control_pi_solve(ctl.h, C.c_size_t(5), C.c_double(1e-7), cost, aa, ctl.opt, 0, None))
        L.valuef_destroy(cost)
        tmp = C.c_void_p(L.c3control_vi_solve(ctl.h, C.c_size_t(1), C.c_double(1e-7), nxt, aa, ctl.opt, 0, None))
        diffs.append(L.valuef_norm2diff(nxt, tmp))
        L.valuef_destroy(nxt)
        cost = tmp
    assert diffs[-1] < diffs[0]
    L.valuef_destroy(cost)
    L.approx_args_free(aa)
    ctl.close()
