"""ctypes binding of oracle/libc3sc_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (c3sc_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_size_p = C.POINTER(C.c_size_t)
c_int_p = C.POINTER(C.c_int)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libc3sc_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libc3sc_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_bellmanrhs.restype = C.c_double
        L.orc_valuef_eval_ind.restype = C.c_double
        L.orc_convert_x_to_ind.restype = C.c_size_t
        L.orc_hashchar.restype = C.c_size_t
        L.orc_hashchar.argtypes = [C.c_size_t, C.c_char_p]
        L.orc_size_t_a_to_char.restype = C.c_char_p
        for name in ("orc_boundary_alloc", "orc_valuef_create", "orc_htable_create", "orc_problem_create",
                     "orc_problem_boundary"):
            getattr(L, name).restype = C.c_void_p
        L.orc_htable_get_element.restype = c_double_p
        L.orc_htable_count.restype = C.c_size_t
        L.orc_problem_xgrid.restype = c_double_p
        L.orc_problem_t.restype = c_double_p
        L.orc_problem_h2.restype = C.c_double
        L.orc_problem_nnode_evals.restype = C.c_size_t
        L.orc_boundary_get_nobs.restype = C.c_size_t
        L.orc_linspace.restype = c_double_p
        _LIB = L
    return _LIB


def dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def sp(a):
    return a.ctypes.data_as(c_size_p) if a is not None else None


def ip(a):
    return a.ctypes.data_as(c_int_p) if a is not None else None


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def usz(a):
    return np.ascontiguousarray(a, dtype=np.uintp)


def ptr_array(arrs):
    """double** from a list of float64 arrays (keeps them alive on the returned object)."""
    arr = (c_double_p * len(arrs))(*[dp(a) for a in arrs])
    arr._keep = arrs
    return arr


# ---------------------------------------------------------------------------------------------
def transition_assemble(dx, du, dw, h2, tvec, drift, ddiff, grad_drift=None, grad_ddiff=None, old=False):
    L = lib()
    prob = np.full(2 * dx + 1, np.nan)
    dt = C.c_double(np.nan)
    fn = L.orc_transition_assemble_old if old else L.orc_transition_assemble
    if grad_drift is None:
        res = fn(C.c_size_t(dx), C.c_size_t(du), C.c_size_t(dw), C.c_double(h2), dp(f64(tvec)), dp(f64(drift)), None,
                 dp(f64(ddiff)), None, dp(prob), None, C.byref(dt), None, None)
        return res, prob, dt.value, None, None
    gp = np.zeros((2 * dx + 1) * du)
    gdt = np.zeros(du)
    space = np.zeros(du)
    res = fn(C.c_size_t(dx), C.c_size_t(du), C.c_size_t(dw), C.c_double(h2), dp(f64(tvec)), dp(f64(drift)),
             dp(f64(grad_drift)), dp(f64(ddiff)), dp(f64(grad_ddiff)), dp(prob), dp(gp), C.byref(dt), dp(gdt),
             dp(space))
    return res, prob, dt.value, gp, gdt


def bellmanrhs(dx, du, stage, discount, prob, dt, cost, stage_grad=None, prob_grad=None, dtgrad=None):
    L = lib()
    if stage_grad is None:
        return L.orc_bellmanrhs(C.c_size_t(dx), C.c_size_t(du), C.c_double(stage), None, C.c_double(discount),
                                dp(f64(prob)), None, C.c_double(dt), None, dp(f64(cost)), None), None
    g = np.zeros(du)
    v = L.orc_bellmanrhs(C.c_size_t(dx), C.c_size_t(du), C.c_double(stage), dp(f64(stage_grad)), C.c_double(discount),
                         dp(f64(prob)), dp(f64(prob_grad)), C.c_double(dt), dp(f64(dtgrad)), dp(f64(cost)), dp(g))
    return v, g


def key_string(arr):
    L = lib()
    buf = C.create_string_buffer(256)
    a = usz(arr)
    L.orc_size_t_a_to_char(sp(a), C.c_size_t(len(a)), buf)
    return buf.value


def hashchar(size, s: bytes):
    return lib().orc_hashchar(C.c_size_t(size), s)


class Boundary:
    def __init__(self, lb, ub):
        self.L = lib()
        self.d = len(lb)
        self.h = C.c_void_p(self.L.orc_boundary_alloc(C.c_size_t(self.d), dp(f64(lb)), dp(f64(ub))))
        self.owned = True

    @classmethod
    def borrowed(cls, handle, d):
        self = cls.__new__(cls)
        self.L = lib()
        self.d = d
        self.h = C.c_void_p(handle)
        self.owned = False
        return self

    def set_type(self, dim, name):
        assert self.L.orc_boundary_external_set_type(self.h, C.c_size_t(dim), name.encode()) == 0

    def add_obstacle(self, center, widths):
        assert self.L.orc_boundary_add_obstacle(self.h, dp(f64(center)), dp(f64(widths))) == 0

    def in_obstacle(self, x):
        return self.L.orc_boundary_in_obstacle(self.h, dp(f64(x)))

    def __del__(self):
        if getattr(self, "owned", False):
            self.L.orc_boundary_free(self.h)


def convert_fiber_to_ind(x, ngrid, xgrid):
    L = lib()
    d = len(ngrid)
    x = f64(x)
    N = x.shape[0]
    fi = np.zeros(d, dtype=np.uintp)
    dv = C.c_size_t(0)
    xs = [f64(g) for g in xgrid]
    res = L.orc_convert_fiber_to_ind(C.c_size_t(d), C.c_size_t(N), dp(x), sp(usz(ngrid)), ptr_array(xs), sp(fi),
                                     C.byref(dv))
    return res, fi, dv.value


def process_fibers_neighbor(fixed_ind, dim_vary, x, ngrid, bound: Boundary):
    L = lib()
    d = len(ngrid)
    N = ngrid[dim_vary]
    absorbed = np.zeros(N, dtype=np.int32)
    nv = np.zeros(2 * N, dtype=np.uintp)
    nf = np.zeros(max(2 * (d - 1), 1), dtype=np.uintp)
    res = L.orc_process_fibers_neighbor(C.c_size_t(d), sp(usz(fixed_ind)), C.c_size_t(dim_vary), dp(f64(x)),
                                        ip(absorbed), sp(nv), sp(nf), sp(usz(ngrid)), bound.h)
    return res, absorbed, nv, nf


class ValueF:
    """Nodal FT value function; cores[m] float64 (N_m, r_m*r_{m+1}) in the reference layout."""

    def __init__(self, ngrid, ranks, cores):
        self.L = lib()
        self.d = len(ngrid)
        self.ngrid = list(ngrid)
        self.ranks = list(ranks)
        self.cores = [f64(c) for c in cores]
        self.h = C.c_void_p(self.L.orc_valuef_create(C.c_size_t(self.d), sp(usz(ngrid)), sp(usz(ranks)),
                                                    ptr_array(self.cores)))

    def eval_ind(self, ind):
        return self.L.orc_valuef_eval_ind(self.h, sp(usz(ind)))

    def eval_fiber_ind_nn(self, fixed_ind, dim_vary, nb_fixed, nb_vary):
        N = self.ngrid[dim_vary]
        out = np.zeros(N * (2 * self.d + 1))
        res = self.L.orc_valuef_eval_fiber_ind_nn(self.h, sp(usz(fixed_ind)), C.c_size_t(dim_vary), sp(usz(nb_fixed)),
                                                  sp(usz(nb_vary)), dp(out))
        assert res == 0
        return out.reshape(N, 2 * self.d + 1)

    def __del__(self):
        self.L.orc_valuef_destroy(self.h)


class Problem:
    """orc_problem built from a c3sc_amd.workloads.Workload (built-in model, brute-force candidates)."""

    def __init__(self, w, cores=None, consistent_ends=False):
        self.L = lib()
        self.w = w
        self.h = C.c_void_p(self.L.orc_problem_create(C.c_size_t(w.dx), C.c_size_t(w.du), C.c_size_t(w.dw),
                                                     dp(f64(w.lb)), dp(f64(w.ub)), sp(usz(w.ngrid)),
                                                     C.c_double(w.discount)))
        b = Boundary.borrowed(self.L.orc_problem_boundary(self.h), w.dx)
        for m, name in enumerate(w.bc_names()):
            b.set_type(m, name)
        for c, wd in w.obstacles:
            b.add_obstacle(c, wd)
        self.bound = b
        if consistent_ends:  # mirrors c3control_set_consistent_ends / c3sc_hip_set_consistent_ends (not the reference's rule)
            self.L.orc_boundary_set_consistent_ends(b.h, C.c_int(1))
        prm = f64(list(w.params) if len(w.params) else [0.0])
        self.L.orc_problem_set_model(self.h, C.c_int(w.model), dp(prm), C.c_size_t(len(w.params)))
        cands = f64(w.cands)
        self.L.orc_problem_set_bruteforce(self.h, C.c_size_t(cands.shape[0]), dp(cands))
        self.vf = None
        if cores is not None:
            self.set_cores(cores)

    def set_cores(self, cores):
        self.vf = ValueF(self.w.ngrid, self.w.ranks, cores)
        self.L.orc_problem_set_value(self.h, self.vf.h)

    def xgrid(self, m):
        p = self.L.orc_problem_xgrid(self.h, C.c_size_t(m))
        return np.ctypeslib.as_array(p, shape=(self.w.ngrid[m],)).copy()

    def h2(self):
        return self.L.orc_problem_h2(self.h)

    def tvec(self):
        p = self.L.orc_problem_t(self.h)
        return np.ctypeslib.as_array(p, shape=(2 * self.w.dx,)).copy()

    def bellman_fibers(self, k, idx, want_absorbed=True):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        F = idx.shape[0]
        N = self.w.ngrid[k]
        out = np.zeros((F, N))
        uidx = np.zeros((F, N), dtype=np.int32)
        ab = np.zeros((F, N), dtype=np.int32) if want_absorbed else None
        res = self.L.orc_bellman_fibers(self.h, C.c_size_t(k), C.c_size_t(F), ip(idx), dp(out), ip(uidx), ip(ab))
        assert res == 0, f"oracle bellman_fibers failed: {res}"
        return out, uidx, ab

    def boundary_handle(self):
        self.L.orc_problem_boundary.restype = C.c_void_p
        return C.c_void_p(self.L.orc_problem_boundary(self.h))

    # ---- policy iteration (bellman_pi, bellman.c:1702-1886)
    def pi_begin(self):
        self.L.orc_problem_pi_begin(self.h)

    def pi_step_begin(self):
        self.L.orc_problem_pi_step_begin(self.h)

    def npol_evals(self):
        self.L.orc_problem_npol_evals.restype = C.c_size_t
        return self.L.orc_problem_npol_evals(self.h)

    def niter_node_evals(self):
        self.L.orc_problem_niter_node_evals.restype = C.c_size_t
        return self.L.orc_problem_niter_node_evals(self.h)

    def policy_fibers(self, policy_vf, k, idx):
        """bellman_pi per fiber: the policy is greedy for policy_vf (a ValueF), the iterate is this problem's value."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        F, N = idx.shape[0], self.w.ngrid[k]
        out = np.zeros((F, N))
        uidx = np.zeros((F, N), dtype=np.int32)
        res = self.L.orc_policy_fibers(self.h, policy_vf.h, C.c_size_t(k), C.c_size_t(F), ip(idx), dp(out), ip(uidx))
        assert res == 0, f"oracle policy_fibers failed: {res}"
        return out, uidx

    def bellman_pi(self, policy_vf, x):
        x = f64(x)
        N = x.shape[0]
        out = np.zeros(N)
        uidx = np.zeros(N, dtype=np.int32)
        res = self.L.orc_bellman_pi(self.h, policy_vf.h, C.c_size_t(N), dp(x), dp(out), ip(uidx))
        assert res == 0
        return out, uidx

    def stencil_fibers(self, k, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        F = idx.shape[0]
        N = self.w.ngrid[k]
        S = 2 * self.w.dx + 1
        out = np.zeros((F, N, S))
        ab = np.zeros((F, N), dtype=np.int32)
        res = self.L.orc_stencil_fibers(self.h, C.c_size_t(k), C.c_size_t(F), ip(idx), dp(out), ip(ab))
        assert res == 0
        return out, ab

    def bellman_vi(self, x, use_memo=True):
        x = f64(x)
        N = x.shape[0]
        out = np.zeros(N)
        uidx = np.zeros(N, dtype=np.int32)
        res = self.L.orc_bellman_vi(self.h, C.c_size_t(N), dp(x), dp(out), ip(uidx), C.c_int(1 if use_memo else 0))
        assert res == 0
        return out, uidx

    def nnode_evals(self):
        return self.L.orc_problem_nnode_evals(self.h)

    def increment_vi_iter(self):
        self.L.orc_problem_increment_vi_iter(self.h)

    def __del__(self):
        self.L.orc_problem_destroy(self.h)
