"""ctypes view of c3sc_amd/host/libc3sc.so -- the C host side that keeps the reference's API names
(include/c3sc/*.h).  Used by tests only; a C program links the library directly."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "c3sc_amd", "host", "libc3sc.so")
_L = None

c_double_p = C.POINTER(C.c_double)
c_size_p = C.POINTER(C.c_size_t)
DYN_FN = C.CFUNCTYPE(C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, C.c_void_p)
STAGE_FN = C.CFUNCTYPE(C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p)
BOUND_FN = C.CFUNCTYPE(C.c_int, C.c_double, c_double_p, c_double_p)
OBS_FN = C.CFUNCTYPE(C.c_int, c_double_p, c_double_p)
FIBER_FN = C.CFUNCTYPE(C.c_int, C.c_size_t, c_double_p, c_double_p, C.c_void_p)


def lib():
    global _L
    if _L is None:
        import torch  # noqa: F401  (same reason as c3sc_amd.engine: one HIP runtime per process)

        L = C.CDLL(PATH)
        for n in ("boundary_alloc", "drift_alloc", "diff_alloc", "htable_create", "workspace_alloc", "mca_param_create",
                  "dp_param_create", "control_params_create", "vi_param_create", "c3control_create", "c3opt_alloc",
                  "valuef_create_nodal", "valuef_copy", "c3control_begin_vi", "approx_args_init", "c3control_begin_pi", "pi_param_create", "c3control_get_boundary"):
            getattr(L, n).restype = C.c_void_p
        L.size_t_a_to_char.restype = C.c_char_p
        L.c3sc_hashchar.restype = C.c_size_t
        L.c3sc_hashchar.argtypes = [C.c_size_t, C.c_char_p]
        L.bellmanrhs.restype = C.c_double
        L.bellmanrhs.argtypes = [C.c_size_t, C.c_size_t, C.c_double, c_double_p, C.c_double, c_double_p, c_double_p,
                                 C.c_double, c_double_p, c_double_p, c_double_p]
        L.transition_assemble.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_double] + [c_double_p] * 10
        L.htable_get_element.restype = c_double_p
        L.c3control_get_xgrid.restype = C.POINTER(c_double_p)
        L.valuef_eval_ind.restype = C.c_double
        L.vi_param_get_nnode_evals.restype = C.c_size_t
        L.pi_param_get_npol_evals.restype = C.c_size_t
        L.pi_param_get_niter_node_evals.restype = C.c_size_t
        L.uniform_stride.restype = C.c_size_t
        L.approx_args_get_cross_tol.restype = C.c_double
        L.approx_args_get_maxrank.restype = C.c_size_t
        _L = L
    return _L


def dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def sp(a):
    return a.ctypes.data_as(c_size_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def usz(a):
    return np.ascontiguousarray(a, dtype=np.uintp)


def ptrs(arrs):
    arr = (c_double_p * len(arrs))(*[dp(a) for a in arrs])
    arr._keep = arrs
    return arr


class Control:
    """c3control_create + problem wiring from a workloads.Workload (device model + optional host callbacks)."""

    def __init__(self, w, callbacks=None, device_model=True, box=None, consistent_ends=False):
        """consistent_ends: False = the reference's literal end-point rule (what the per-fiber parity tests compare with the
        literal oracle), True = the library's solver default (c3control_set_consistent_ends), None = leave the default."""
        L = lib()
        self.L, self.w = L, w
        self._lb, self._ub, self._ng = f64(w.lb), f64(w.ub), usz(w.ngrid)
        self.h = C.c_void_p(L.c3control_create(C.c_size_t(w.dx), C.c_size_t(w.du), C.c_size_t(w.dw), dp(self._lb), dp(self._ub),
                                               sp(self._ng), C.c_double(w.discount)))
        if consistent_ends is not None:
            L.c3control_set_consistent_ends(self.h, C.c_int(1 if consistent_ends else 0))
        for m, name in enumerate(w.bc_names()):
            L.c3control_set_external_boundary(self.h, C.c_size_t(m), name.encode())
        for cen, wid in w.obstacles:
            L.c3control_add_obstacle(self.h, dp(f64(cen)), dp(f64(wid)))
        prm = f64(list(w.params) if len(w.params) else [0.0])
        if device_model:
            L.c3control_set_device_model(self.h, C.c_int(w.model), dp(prm), C.c_size_t(len(w.params)))
        self._cb = callbacks
        if callbacks is not None:
            b, s, st, bd, ob = callbacks
            L.c3control_add_drift(self.h, b, None)
            L.c3control_add_diff(self.h, s, None)
            L.c3control_add_stagecost(self.h, st)
            L.c3control_add_boundcost(self.h, bd)
            L.c3control_add_obscost(self.h, ob)
        if box is None:
            self.opt = C.c_void_p(L.c3opt_alloc(C.c_int(3), C.c_size_t(w.du)))  # BRUTEFORCE
            cands = f64(w.cands)
            L.c3opt_set_brute_force_vals(self.opt, C.c_size_t(cands.shape[0]), dp(cands))
        else:  # the examples' set-up for continuous controls (e.g. tprob_test.c:2291-2299): BFGS + box bounds
            lb, ub = f64(box[0]), f64(box[1])
            self.opt = C.c_void_p(L.c3opt_alloc(C.c_int(0), C.c_size_t(w.du)))
            L.c3opt_add_lb(self.opt, dp(lb))
            L.c3opt_add_ub(self.opt, dp(ub))
            L.c3opt_set_relftol(self.opt, C.c_double(1e-8))
            L.c3opt_set_gtol(self.opt, C.c_double(1e-30))
            L.c3opt_ls_set_maxiter(self.opt, C.c_size_t(10))
            L.c3opt_set_verbose(self.opt, C.c_int(0))
            L.c3opt_set_maxiter(self.opt, C.c_size_t(10))

    def xgrid(self):
        pp = self.L.c3control_get_xgrid(self.h)
        return [np.ctypeslib.as_array(pp[m], shape=(self.w.ngrid[m],)).copy() for m in range(self.w.dx)]

    def bound(self):
        return C.c_void_p(self.L.c3control_get_boundary(self.h))

    def valuef(self, cores):
        cs = [f64(c) for c in cores]
        return C.c_void_p(self.L.valuef_create_nodal(C.c_size_t(self.w.dx), sp(usz(self.w.ngrid)), sp(usz(self.w.ranks)), ptrs(cs)))

    def begin_vi(self, vf):
        return C.c_void_p(self.L.c3control_begin_vi(self.h, vf, self.opt))

    def end_vi(self, vi):
        n = C.c_size_t(0)
        self.L.c3control_end_vi(self.h, vi, C.byref(n))
        return n.value

    def bellman_vi(self, vi, x):
        x = f64(x)
        out = np.zeros(x.shape[0])
        rc = self.L.bellman_vi(C.c_size_t(x.shape[0]), dp(x), dp(out), vi)
        assert rc == 0
        return out

    def bellman_vi_batch(self, vi, x):
        x = f64(x)  # (F, N, dx)
        F, N = x.shape[0], x.shape[1]
        out = np.zeros((F, N))
        rc = self.L.bellman_vi_batch(C.c_size_t(F), C.c_size_t(N), dp(x), dp(out), vi)
        assert rc == 0
        return out

    # ---- policy iteration (bellman_pi)
    def begin_pi(self, policy_vf):
        return C.c_void_p(self.L.c3control_begin_pi(self.h, policy_vf))

    def begin_pi_step(self, pi, vf):
        self.L.c3control_begin_pi_step(self.h, pi, vf, self.opt)

    def end_pi_step(self, pi):
        n = C.c_size_t(0)
        self.L.c3control_end_pi_step(self.h, pi, C.byref(n))
        return n.value

    def bellman_pi(self, pi, x):
        x = f64(x)
        out = np.zeros(x.shape[0])
        rc = self.L.bellman_pi(C.c_size_t(x.shape[0]), dp(x), dp(out), pi)
        assert rc == 0
        return out

    def bellman_pi_batch(self, pi, x):
        x = f64(x)
        F, N = x.shape[0], x.shape[1]
        out = np.zeros((F, N))
        rc = self.L.bellman_pi_batch(C.c_size_t(F), C.c_size_t(N), dp(x), dp(out), pi)
        assert rc == 0
        return out

    def nnode_evals(self, vi):
        return self.L.vi_param_get_nnode_evals(vi)

    def close(self):
        self.L.c3opt_free(self.opt)
        self.L.c3control_destroy(self.h)
