"""ctypes binding of libc3sc_hip.so (include/c3sc_hip.h) + a thin host-side engine object.

PyTorch is used only as plumbing: device buffers (torch tensors' data_ptr()), streams and
torch.distributed.  The compute path is the HIP library; if it is missing this module raises --
there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libc3sc_hip.so")
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_size_p = C.POINTER(C.c_size_t)
c_int_p = C.POINTER(C.c_int)
c_i32_p = C.POINTER(C.c_int32)

# every symbol include/c3sc_hip.h declares (the CPU test-suite checks they are all exported)
EXPORTS = [
    "c3sc_hip_ctx_create", "c3sc_hip_ctx_destroy", "c3sc_hip_last_error", "c3sc_hip_device_count", "c3sc_hip_max_rank",
    "c3sc_hip_set_grid", "c3sc_hip_set_boundary", "c3sc_hip_set_consistent_ends", "c3sc_hip_get_consistent_ends", "c3sc_hip_cross_setup", "c3sc_hip_cross_iteration", "c3sc_hip_cross_iteration_streamed", "c3sc_hip_cross_wait_core", "c3sc_hip_cross_iteration_pi", "c3sc_hip_cross_confirm", "c3sc_hip_bellman_fibers_all", "c3sc_hip_policy_fibers_all", "c3sc_hip_cross_speculate", "c3sc_hip_comm_unique_id", "c3sc_hip_comm_create", "c3sc_hip_comm_destroy", "c3sc_hip_comm_world", "c3sc_hip_comm_rank", "c3sc_hip_comm_allgather", "c3sc_hip_cross_set_comm", "c3sc_hip_comm_exchange", "c3sc_hip_cross_options", "c3sc_hip_cross_grow_memo", "c3sc_hip_cross_fetch", "c3sc_hip_cross_free", "c3sc_hip_set_mca", "c3sc_hip_set_model",
    "c3sc_hip_set_controls", "c3sc_hip_upload_value", "c3sc_hip_upload_value_device", "c3sc_hip_set_variant",
    "c3sc_hip_bellman_fibers", "c3sc_hip_bellman_fibers_tables", "c3sc_hip_bellman_fibers_tables_host", "c3sc_hip_stencil_fibers", "c3sc_hip_bellman_fibers_host",
    "c3sc_hip_policy_fibers", "c3sc_hip_policy_fibers_host", "c3sc_hip_policy_fibers_tables", "c3sc_hip_policy_fibers_tables_host",
    "c3sc_hip_set_control_box", "c3sc_hip_bellman_fibers_box", "c3sc_hip_bellman_fibers_box_host", "c3sc_hip_policy_fibers_box",
    "c3sc_hip_policy_fibers_box_host",
    "c3sc_hip_stencil_fibers_host", "c3sc_hip_stencil_fibers_nb", "c3sc_hip_stencil_fibers_nb_host", "c3sc_hip_sync", "c3sc_hip_get_status", "c3sc_hip_last_kernel",
    "c3sc_hip_debug_read", "c3sc_hip_launch_count", "c3sc_hip_timer_start", "c3sc_hip_timer_stop", "c3sc_hip_peak_fma_f64", "c3sc_hip_peak_mfma_f64",
]

VARIANT_AUTO, VARIANT_FIBER_PER_WAVE, VARIANT_FIBER_PER_LANE, VARIANT_FIBER_PAIR, VARIANT_FIBER_QUAD = 0, 1, 2, 3, 4


class C3scHipError(RuntimeError):
    pass


def load_library():
    """Load libc3sc_hip.so or fail loudly (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise C3scHipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C c3sc_amd/csrc`.  There is no CPU fallback.")
        # PyTorch bundles its own libamdhip64.so.7; load it FIRST so that this library binds to the same
        # HIP runtime instance (two runtimes in one process cannot both own the GPU).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.c3sc_hip_last_error.restype = C.c_char_p
        L.c3sc_hip_last_error.argtypes = [C.c_void_p]
        L.c3sc_hip_last_kernel.restype = C.c_char_p
        L.c3sc_hip_last_kernel.argtypes = [C.c_void_p]
        L.c3sc_hip_ctx_destroy.restype = None
        L.c3sc_hip_ctx_destroy.argtypes = [C.c_void_p]
        L.c3sc_hip_bellman_fibers.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]
        L.c3sc_hip_stencil_fibers.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
        L.c3sc_hip_bellman_fibers_host.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p]
        L.c3sc_hip_stencil_fibers_host.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.c3sc_hip_policy_fibers.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
        L.c3sc_hip_policy_fibers_host.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p]
        L.c3sc_hip_stencil_fibers_nb_host.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_void_p]
        L.c3sc_hip_bellman_fibers_tables_host.argtypes = [C.c_void_p, C.c_int, C.c_size_t] + [C.c_void_p] * 6
        L.c3sc_hip_launch_count.restype = C.c_ulonglong
        L.c3sc_hip_sync.argtypes = [C.c_void_p, C.c_void_p]
        L.c3sc_hip_timer_start.argtypes = [C.c_void_p, C.c_void_p]
        L.c3sc_hip_timer_stop.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.c3sc_hip_upload_value_device.argtypes = [C.c_void_p, c_size_p, C.POINTER(C.c_void_p), C.c_void_p]
        _LIB = L
    return _LIB


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr_array(arrs):
    arr = (c_double_p * len(arrs))(*[a.ctypes.data_as(c_double_p) for a in arrs])
    arr._keep = arrs
    return arr


class BellmanEngine:
    """Device-resident Bellman-backup problem: the HIP counterpart of C3Control + VIparam
    (src/bellman.c:1942-1999, 1132-1180) for one GPU."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.c3sc_hip_ctx_create(C.c_int(device), C.byref(h))
        if rc != 0:
            raise C3scHipError(f"c3sc_hip_ctx_create(device={device}) failed with code {rc} (no usable HIP device?)")
        self.h = h
        self.device = device
        self.w = None

    def close(self):
        if getattr(self, "h", None):
            self.L.c3sc_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.L.c3sc_hip_last_error(self.h)
            raise C3scHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

    # ------------------------------------------------------------------ problem description
    def set_grid(self, ngrid: Sequence[int], xgrid: Sequence[np.ndarray]):
        ng = np.ascontiguousarray(ngrid, dtype=np.uintp)
        xs = [_f64(g) for g in xgrid]
        self._chk(self.L.c3sc_hip_set_grid(self.h, C.c_int(len(ng)), ng.ctypes.data_as(c_size_p), _ptr_array(xs)),
                  "set_grid")
        self.ngrid = [int(n) for n in ngrid]
        self.d = len(self.ngrid)

    def set_boundary(self, bctype: Sequence[int], obstacles=()):
        bc = np.ascontiguousarray(bctype, dtype=np.int32)
        lb = _f64([[c - wd / 2.0 for c, wd in zip(cen, wid)] for cen, wid in obstacles]).reshape(-1)
        ub = _f64([[c + wd / 2.0 for c, wd in zip(cen, wid)] for cen, wid in obstacles]).reshape(-1)
        self._chk(self.L.c3sc_hip_set_boundary(self.h, bc.ctypes.data_as(c_int_p), C.c_int(len(obstacles)),
                                               lb.ctypes.data_as(c_double_p), ub.ctypes.data_as(c_double_p)),
                  "set_boundary")

    def set_mca(self, h2: float, t: Sequence[float], discount: float):
        tv = _f64(t)
        self._chk(self.L.c3sc_hip_set_mca(self.h, C.c_double(h2), tv.ctypes.data_as(c_double_p), C.c_double(discount)),
                  "set_mca")

    def set_model(self, model: int, params: Sequence[float] = ()):
        p = _f64(list(params) if len(params) else [0.0])
        self._chk(self.L.c3sc_hip_set_model(self.h, C.c_int(model), p.ctypes.data_as(c_double_p), C.c_int(len(params))),
                  "set_model")

    def set_controls(self, cands: np.ndarray):
        cd = _f64(cands)
        self._chk(self.L.c3sc_hip_set_controls(self.h, C.c_int(cd.shape[0]), C.c_int(cd.shape[1]),
                                               cd.ctypes.data_as(c_double_p)), "set_controls")

    def set_variant(self, variant: int):
        self._chk(self.L.c3sc_hip_set_variant(self.h, C.c_int(variant)), "set_variant")

    def set_consistent_ends(self, on: bool):
        """Not the reference's rule (default off): end points of reflecting / periodic fibers keep their absorbed flags."""
        self._chk(self.L.c3sc_hip_set_consistent_ends(self.h, C.c_int(1 if on else 0)), "set_consistent_ends")

    def upload_value(self, ranks: Sequence[int], cores: Sequence[np.ndarray]):
        rk = np.ascontiguousarray(ranks, dtype=np.uintp)
        cs = [_f64(c) for c in cores]
        self._chk(self.L.c3sc_hip_upload_value(self.h, rk.ctypes.data_as(c_size_p), _ptr_array(cs)), "upload_value")
        self.ranks = [int(r) for r in ranks]

    def upload_value_device(self, ranks: Sequence[int], core_tensors, stream_ptr: int = 0):
        """cores already on this GPU (torch float64 tensors in the reference layout)."""
        rk = np.ascontiguousarray(ranks, dtype=np.uintp)
        ptrs = (C.c_void_p * len(core_tensors))(*[C.c_void_p(t.data_ptr()) for t in core_tensors])
        self._chk(self.L.c3sc_hip_upload_value_device(self.h, rk.ctypes.data_as(c_size_p), ptrs, C.c_void_p(stream_ptr)),
                  "upload_value_device")
        self.ranks = [int(r) for r in ranks]

    def configure(self, w, cores=None):
        """Everything from a c3sc_amd.workloads.Workload.  Grid constants follow c3control_create /
        mca_add_grid_refs (src/bellman.c:1972-1986, 171-188)."""
        xg = w.xgrid()
        self.set_grid(w.ngrid, xg)
        self.set_boundary(w.bc, w.obstacles)
        hs = [g[1] - g[0] for g in xg]
        hmin = w.ub[0] - w.lb[0]
        for h in hs:
            if h < hmin:
                hmin = h
        h2 = hmin * hmin
        t = []
        for h in hs:
            t0 = h2 / h
            t += [t0, t0 / h]
        self.set_mca(h2, t, w.discount)
        self.set_model(w.model, w.params)
        self.set_controls(w.cands)
        self.w = w
        if cores is not None:
            self.upload_value(w.ranks, cores)

    # ------------------------------------------------------------------ the hot path
    def bellman_fibers(self, k: int, idx_t, out_t=None, uidx_t=None, absorbed_t=None, stream_ptr: Optional[int] = None):
        """Device API: idx_t int32 CUDA tensor (F, d); returns/reuses float64 CUDA tensor (F, N_k)."""
        import torch

        assert idx_t.is_cuda and idx_t.dtype == torch.int32 and idx_t.is_contiguous()
        F = idx_t.shape[0]
        N = self.ngrid[k]
        if out_t is None:
            out_t = torch.empty((F, N), dtype=torch.float64, device=idx_t.device)
        if stream_ptr is None:
            stream_ptr = torch.cuda.current_stream(idx_t.device).cuda_stream
        self._chk(self.L.c3sc_hip_bellman_fibers(self.h, k, F, idx_t.data_ptr(), out_t.data_ptr(),
                                                 uidx_t.data_ptr() if uidx_t is not None else None,
                                                 absorbed_t.data_ptr() if absorbed_t is not None else None,
                                                 stream_ptr), "bellman_fibers")
        return out_t

    def bellman_fibers_all(self, ks, idx_ts, out_ts, stream_ptr: Optional[int] = None, policy_ts=None):
        """Device API, several varying dimensions of one batch in one call (c3sc_hip_bellman_fibers_all / _policy_fibers_all): one
        launch where a fused instantiation exists.  idx_ts[s]: int32 CUDA tensor (F_s, d); out_ts[s]: float64 (F_s, N_ks[s])."""
        import torch

        nk = len(ks)
        KS = (C.c_int * nk)(*[int(k) for k in ks])
        FS = (C.c_size_t * nk)(*[int(t.shape[0]) for t in idx_ts])
        IDX = (C.c_void_p * nk)(*[t.data_ptr() for t in idx_ts])
        OUT = (C.c_void_p * nk)(*[t.data_ptr() for t in out_ts])
        if stream_ptr is None:
            stream_ptr = torch.cuda.current_stream(idx_ts[0].device).cuda_stream
        if policy_ts is None:
            self._chk(self.L.c3sc_hip_bellman_fibers_all(self.h, nk, KS, FS, IDX, OUT, None, None, C.c_void_p(stream_ptr)), "bellman_fibers_all")
        else:
            POL = (C.c_void_p * nk)(*[t.data_ptr() for t in policy_ts])
            self._chk(self.L.c3sc_hip_policy_fibers_all(self.h, nk, KS, FS, IDX, POL, OUT, None, C.c_void_p(stream_ptr)), "policy_fibers_all")
        return out_ts

    def stencil_fibers(self, k: int, idx_t, costs_t=None, absorbed_t=None, stream_ptr: Optional[int] = None):
        import torch

        assert idx_t.is_cuda and idx_t.dtype == torch.int32 and idx_t.is_contiguous()
        F = idx_t.shape[0]
        N = self.ngrid[k]
        if costs_t is None:
            costs_t = torch.empty((F, N, 2 * self.d + 1), dtype=torch.float64, device=idx_t.device)
        if stream_ptr is None:
            stream_ptr = torch.cuda.current_stream(idx_t.device).cuda_stream
        self._chk(self.L.c3sc_hip_stencil_fibers(self.h, k, F, idx_t.data_ptr(), costs_t.data_ptr(),
                                                 absorbed_t.data_ptr() if absorbed_t is not None else None,
                                                 stream_ptr), "stencil_fibers")
        return costs_t

    def bellman_fibers_host(self, k: int, idx: np.ndarray, want_uidx=True, want_absorbed=True):
        """Host-buffer API (what the C facade's bellman_vi uses): numpy in, numpy out, synchronous."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        F, N = idx.shape[0], self.ngrid[k]
        out = np.empty((F, N))
        ui = np.empty((F, N), dtype=np.int32) if want_uidx else None
        ab = np.empty((F, N), dtype=np.int32) if want_absorbed else None
        self._chk(self.L.c3sc_hip_bellman_fibers_host(self.h, k, F, idx.ctypes.data, out.ctypes.data,
                                                      ui.ctypes.data if ui is not None else None,
                                                      ab.ctypes.data if ab is not None else None), "bellman_fibers_host")
        return out, ui, ab

    def policy_fibers_host(self, k: int, idx: np.ndarray, policy: np.ndarray):
        """Policy evaluation (batched bellman_pi): apply candidate policy[f, j] at every node."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        policy = np.ascontiguousarray(policy, dtype=np.int32)
        F, N = idx.shape[0], self.ngrid[k]
        assert policy.shape == (F, N)
        out = np.empty((F, N))
        ab = np.empty((F, N), dtype=np.int32)
        self._chk(self.L.c3sc_hip_policy_fibers_host(self.h, k, F, idx.ctypes.data, policy.ctypes.data, out.ctypes.data,
                                                     ab.ctypes.data), "policy_fibers_host")
        return out, ab

    # ---- continuous controls in a box
    def set_control_box(self, lb, ub, grid=33, polish=2):
        lb, ub = _f64(lb), _f64(ub)
        self.box_du = len(lb)
        self._chk(self.L.c3sc_hip_set_control_box(self.h, len(lb), lb.ctypes.data_as(c_double_p), ub.ctypes.data_as(c_double_p),
                                                  int(grid), int(polish)), "set_control_box")

    def bellman_fibers_box_host(self, k: int, idx: np.ndarray):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        F, N = idx.shape[0], self.ngrid[k]
        out, uo, ab = np.empty((F, N)), np.empty((F, N, self.box_du)), np.empty((F, N), dtype=np.int32)
        self._chk(self.L.c3sc_hip_bellman_fibers_box_host(self.h, k, C.c_size_t(F), C.c_void_p(idx.ctypes.data), C.c_void_p(out.ctypes.data),
                                                          C.c_void_p(uo.ctypes.data), C.c_void_p(ab.ctypes.data)), "bellman_fibers_box_host")
        return out, uo, ab

    def policy_fibers_box_host(self, k: int, idx: np.ndarray, policy_u: np.ndarray):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        pu = _f64(policy_u)
        F, N = idx.shape[0], self.ngrid[k]
        out, ab = np.empty((F, N)), np.empty((F, N), dtype=np.int32)
        self._chk(self.L.c3sc_hip_policy_fibers_box_host(self.h, k, C.c_size_t(F), C.c_void_p(idx.ctypes.data), C.c_void_p(pu.ctypes.data),
                                                         C.c_void_p(out.ctypes.data), C.c_void_p(ab.ctypes.data)), "policy_fibers_box_host")
        return out, ab

    def bellman_fibers_tables_host(self, k: int, idx: np.ndarray, tables: np.ndarray, costs2: np.ndarray):
        """Universal path: tables (F, N, U, 2d+1) = host-evaluated (drift, diag sigma, stage), costs2 (F, N, 2)."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        tables = _f64(tables)
        costs2 = _f64(costs2)
        F, N = idx.shape[0], self.ngrid[k]
        out = np.empty((F, N))
        ui = np.empty((F, N), dtype=np.int32)
        ab = np.empty((F, N), dtype=np.int32)
        self._chk(self.L.c3sc_hip_bellman_fibers_tables_host(self.h, k, F, idx.ctypes.data, tables.ctypes.data,
                                                             costs2.ctypes.data, out.ctypes.data, ui.ctypes.data,
                                                             ab.ctypes.data), "bellman_fibers_tables_host")
        return out, ui, ab

    def stencil_fibers_host(self, k: int, idx: np.ndarray, nb_fixed=None, nb_vary=None):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        F, N = idx.shape[0], self.ngrid[k]
        costs = np.empty((F, N, 2 * self.d + 1))
        ab = np.empty((F, N), dtype=np.int32)
        nf = np.ascontiguousarray(nb_fixed, dtype=np.int32) if nb_fixed is not None else None
        nv = np.ascontiguousarray(nb_vary, dtype=np.int32) if nb_vary is not None else None
        self._chk(self.L.c3sc_hip_stencil_fibers_nb_host(self.h, k, F, idx.ctypes.data,
                                                         nf.ctypes.data if nf is not None else None,
                                                         nv.ctypes.data if nv is not None else None,
                                                         costs.ctypes.data, ab.ctypes.data), "stencil_fibers_nb_host")
        return costs, ab

    # ------------------------------------------------------------------ misc
    def sync(self, stream_ptr: int = 0):
        self._chk(self.L.c3sc_hip_sync(self.h, C.c_void_p(stream_ptr)), "sync")

    def status(self, clear=True) -> int:
        fl = C.c_uint(0)
        self._chk(self.L.c3sc_hip_get_status(self.h, C.byref(fl), C.c_int(1 if clear else 0)), "get_status")
        return fl.value

    def last_kernel(self) -> str:
        return self.L.c3sc_hip_last_kernel(self.h).decode()

    def timer_start(self, stream_ptr: int = 0):
        self._chk(self.L.c3sc_hip_timer_start(self.h, C.c_void_p(stream_ptr)), "timer_start")

    def timer_stop(self, stream_ptr: int = 0) -> float:
        ms = C.c_float(0)
        self._chk(self.L.c3sc_hip_timer_stop(self.h, C.c_void_p(stream_ptr), C.byref(ms)), "timer_stop")
        return ms.value

    def debug_read(self, n: int) -> np.ndarray:
        buf = np.zeros(n, dtype=np.uint64)
        self._chk(self.L.c3sc_hip_debug_read(self.h, C.c_void_p(buf.ctypes.data), C.c_size_t(n)), "debug_read")
        return buf

    def peak_fma_f64(self) -> float:
        v = C.c_double(0)
        self._chk(self.L.c3sc_hip_peak_fma_f64(self.h, C.byref(v)), "peak_fma_f64")
        return v.value

    def peak_mfma_f64(self) -> float:
        v = C.c_double(0)
        self._chk(self.L.c3sc_hip_peak_mfma_f64(self.h, C.byref(v)), "peak_mfma_f64")
        return v.value
