// models.hpp -- device problem models for the Bellman-backup kernels (gfx950).
//
// The reference describes a problem with five host callbacks (drift, diffusion, stage cost,
// boundary cost, obstacle cost: src/dynamics.c:127-139,224-239, src/bellman.c:215-217).  A
// kernel cannot call host function pointers, so each benchmark problem is restated as a
// compile-time functor that is inlined into the node loop.  Only the DIAGONAL of the diffusion
// matrix is produced because that is all transition_assemble reads (src/nodeutil.c:294).
//
// Transcendentals never run on the device: every model lists the univariate functions of a grid
// coordinate it needs (cos(theta_i), tan(delta_i), ...) as TABLES over that dimension's nodes and the
// functions of a control candidate (tan(u0)) as per-candidate FEATURES; the host fills both with libm
// (c3sc_hip.hip: model_table_value / model_cand_feature), so the kernels read the very doubles the
// reference's callbacks would compute and no polynomial constants occupy VGPRs.
//
// Every model has
//   D, DU                       state / control dimension
//   NTAB, tab_dim(t)            number of tables and the dimension each one is indexed by
//   NCF                         number of per-candidate features
//   UDEP_MASK, UCONST_MASK      dims whose drift/diffusion depend on the control; those of them that depend on
//                               nothing else, so their transition rates are constants of the candidate
//   Node                        per-node invariants computed once per node, not per control
//   prep(prm, x, tv, node)      tv[t] = value of table t at this node (the kernel looks it up)
//   drift(prm, node, x, u, cf, b)   b[D]; cf = this candidate's features
//   sigma(prm, x, u, s)         s[D] diagonal of the diffusion
//   stage(prm, x, u)            stage cost
//   boundcost(prm, x), obscost(prm, x)
#pragma once
#include <hip/hip_runtime.h>

namespace c3sc {

// examples/dubinscar_new/dubinscar.c:40-121
struct Dubins3D {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = 3, DU = 1;
    static constexpr int NTAB = 2, NCF = 0; // tables: cos(x2), sin(x2)
    static constexpr unsigned UDEP_MASK = 1u << 2; // dims whose drift/diffusion depend on the control
    static constexpr unsigned UCONST_MASK = 1u << 2; // ... and on nothing else (no state): rates are per-candidate constants
    static constexpr bool STAGE_UDEP = false;
    __host__ __device__ static constexpr int tab_dim(int) { return 2; }
    struct Node { double c, s; };
    __device__ static inline void prep(const double *, const double (&)[D], const double (&tv)[2], Node &n)
    {
        n.c = tv[0];
        n.s = tv[1];
    }
    __device__ static inline void drift(const double *, const Node &n, const double (&)[D], const double *u, const double *,
                                        double (&b)[D])
    {
        b[0] = n.c; b[1] = n.s; b[2] = u[0];
    }
    __device__ static inline void sigma(const double *, const double (&)[D], const double *, double (&s)[D])
    {
        s[0] = 1e0; s[1] = 1e0; s[2] = 1e-2;
    }
    __device__ static inline double stage(const double *, const double (&)[D], const double *) { return 1.0; }
    __device__ static inline double boundcost(const double *, const double (&)[D]) { return 10.0; }
    __device__ static inline double obscost(const double *, const double (&)[D]) { return 0.0; }
};

// examples/skidding_car/scar.c:40-169 (order = {0,1,2,3})
struct Scar4D {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = 4, DU = 2;
    static constexpr int NTAB = 3, NCF = 1; // tables: cos(x2), sin(x2), speed factor of x3; candidate feature: tan(u0)
    static constexpr unsigned UDEP_MASK = (1u << 2) | (1u << 3);
    static constexpr unsigned UCONST_MASK = 1u << 3; // b[2] also depends on the speed
    static constexpr bool STAGE_UDEP = false;
    __host__ __device__ static constexpr int tab_dim(int t) { return t == 2 ? 3 : 2; }
    struct Node { double vc, vs, pre; };
    __device__ static inline void prep(const double *, const double (&x)[D], const double (&tv)[3], Node &n)
    {
        const double speed = x[3];
        n.pre = tv[2]; // (1 / (1 + speed/vcar)) * (speed / L), L = 0.2, vcar = 8: tabulated over dim 3 on the host
        n.vc = speed * tv[0];
        n.vs = speed * tv[1];
    }
    __device__ static inline void drift(const double *, const Node &n, const double (&)[D], const double *u, const double *cf,
                                        double (&b)[D])
    {
        b[0] = n.vc; b[1] = n.vs; b[2] = n.pre * cf[0]; b[3] = 2.0 * u[1];
    }
    __device__ static inline void sigma(const double *, const double (&)[D], const double *, double (&s)[D])
    {
        s[0] = 1.0; s[1] = 1.0; s[2] = 1e-2; s[3] = 1e-2;
    }
    __device__ static inline double stage(const double *, const double (&x)[D], const double *)
    {
        return 1.0 + x[0] * x[0] + x[1] * x[1];
    }
    __device__ static inline double boundcost(const double *, const double (&)[D]) { return 10.0; }
    __device__ static inline double obscost(const double *, const double (&)[D]) { return 0.0; }
};

// synthetic 7-D car: state (x, y, theta, v, omega, delta, a), controls (ddelta, da)
struct Car7D {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = 7, DU = 2;
    static constexpr int NTAB = 4, NCF = 0; // tables: cos(x2), sin(x2), tan(x5), v / (0.2 (1 + v/8)) over x3
    static constexpr unsigned UDEP_MASK = (1u << 5) | (1u << 6);
    static constexpr unsigned UCONST_MASK = UDEP_MASK;
    static constexpr bool STAGE_UDEP = false;
    __host__ __device__ static constexpr int tab_dim(int t) { return t == 2 ? 5 : (t == 3 ? 3 : 2); }
    struct Node { double b0, b1, b4; };
    __device__ static inline void prep(const double *, const double (&x)[D], const double (&tv)[4], Node &n)
    {
        const double v = x[3], om = x[4];
        n.b0 = v * tv[0];
        n.b1 = v * tv[1];
        n.b4 = (tv[3] * tv[2] - om) / 0.5; // tv[3] = v / (0.2 (1 + v/8)), tabulated on the host
    }
    __device__ static inline void drift(const double *, const Node &n, const double (&x)[D], const double *u, const double *,
                                        double (&b)[D])
    {
        b[0] = n.b0; b[1] = n.b1; b[2] = x[4]; b[3] = 2.0 * x[6]; b[4] = n.b4; b[5] = u[0]; b[6] = u[1];
    }
    __device__ static inline void sigma(const double *, const double (&)[D], const double *, double (&s)[D])
    {
        s[0] = 1.0; s[1] = 1.0;
#pragma unroll
        for (int i = 2; i < D; i++) s[i] = 1e-2;
    }
    __device__ static inline double stage(const double *, const double (&x)[D], const double *)
    {
        return 1.0 + x[0] * x[0] + x[1] * x[1];
    }
    __device__ static inline double boundcost(const double *, const double (&)[D]) { return 10.0; }
    __device__ static inline double obscost(const double *, const double (&)[D]) { return 0.0; }
};

// examples/lqgnd/lqgnd.c:80-198 (dim = 2 is examples/lqg2d_new/lqg2d.c:72-153); prm = {dim, sig_even, sig_odd}
template <int DIM>
struct LqgNd {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = DIM, DU = DIM / 2;
    static constexpr int NTAB = 0, NCF = 0;
    static constexpr unsigned UDEP_MASK = 0xAAAAAAAAu & ((1u << DIM) - 1u); // odd dims are driven by a control
    static constexpr unsigned UCONST_MASK = UDEP_MASK;
    static constexpr bool STAGE_UDEP = true;
    __host__ __device__ static constexpr int tab_dim(int) { return 0; }
    struct Node {};
    __device__ static inline void prep(const double *, const double (&)[D], const double (&)[1], Node &) {}
    __device__ static inline void drift(const double *, const Node &, const double (&x)[D], const double *u, const double *,
                                        double (&b)[D])
    {
#pragma unroll
        for (int i = 0; i < D; i++) b[i] = ((i % 2) == 0) ? x[(i + 1 < D) ? i + 1 : i] : u[i / 2];
    }
    __device__ static inline void sigma(const double *prm, const double (&)[D], const double *, double (&s)[D])
    {
#pragma unroll
        for (int i = 0; i < D; i++) s[i] = ((i % 2) == 0) ? prm[1] : prm[2];
    }
    __device__ static inline double stage(const double *, const double (&x)[D], const double *u)
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < D; i++) s += x[i] * x[i];
#pragma unroll
        for (int i = 0; i < DU; i++) s += u[i] * u[i];
        return s;
    }
    // stage = stage_x(x) + stage_u(u): the control's share is a constant of the candidate (kept in the candidate table)
    static constexpr bool STAGE_USEP = true;
    __device__ static inline double stage_x(const double *, const double (&x)[D])
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < D; i++) s += x[i] * x[i];
        return s;
    }
    __device__ static inline double stage_u(const double *, const double *u)
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < DU; i++) s += u[i] * u[i];
        return s;
    }
    __device__ static inline double boundcost(const double *, const double (&)[D]) { return 100.0; }
    __device__ static inline double obscost(const double *, const double (&)[D]) { return 0.0; }
};

// examples/double_int/double_int.c:80-157; prm = {dim, sig, sig_last, stage_mode}
// stage_mode 0: stage = 1 (double_int.c:126); 1: stage = sum x_i^2 (synthetic quad10d)
template <int DIM>
struct Chain {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = DIM, DU = 1;
    static constexpr int NTAB = 0, NCF = 0;
    static constexpr unsigned UDEP_MASK = 1u << (DIM - 1);
    static constexpr unsigned UCONST_MASK = UDEP_MASK;
    static constexpr bool STAGE_UDEP = false;
    __host__ __device__ static constexpr int tab_dim(int) { return 0; }
    struct Node {};
    __device__ static inline void prep(const double *, const double (&)[D], const double (&)[1], Node &) {}
    __device__ static inline void drift(const double *, const Node &, const double (&x)[D], const double *u, const double *,
                                        double (&b)[D])
    {
#pragma unroll
        for (int i = 0; i < D - 1; i++) b[i] = x[i + 1];
        b[D - 1] = u[0];
    }
    __device__ static inline void sigma(const double *prm, const double (&)[D], const double *, double (&s)[D])
    {
#pragma unroll
        for (int i = 0; i < D - 1; i++) s[i] = prm[1];
        s[D - 1] = prm[2];
    }
    __device__ static inline double stage(const double *prm, const double (&x)[D], const double *)
    {
        if (prm[3] == 0.0) return 1.0;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < D; i++) s += x[i] * x[i];
        return s;
    }
    __device__ static inline double boundcost(const double *, const double (&)[D]) { return 1000.0; }
    __device__ static inline double obscost(const double *, const double (&)[D]) { return 0.0; }
};

// examples/rossler/rossler.c:80-157 (Roessler attractor with a control on the second equation; a = b = 0.1, c = 14);
// prm = {3, sig, sig_last} like the chain (rossler.c:104-116); stage = 100 |x|^2 + u^2 (:137-149)
struct Rossler3D {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = 3, DU = 1;
    static constexpr int NTAB = 0, NCF = 0;
    static constexpr unsigned UDEP_MASK = 1u << 1;
    static constexpr unsigned UCONST_MASK = 0; // the controlled equation also depends on the state
    static constexpr bool STAGE_UDEP = true;
    __host__ __device__ static constexpr int tab_dim(int) { return 0; }
    struct Node {};
    __device__ static inline void prep(const double *, const double (&)[D], const double (&)[1], Node &) {}
    __device__ static inline void drift(const double *, const Node &, const double (&x)[D], const double *u, const double *,
                                        double (&b)[D])
    {
        b[0] = -x[1] - x[2];
        b[1] = x[0] + 0.1 * x[1] + u[0];
        b[2] = 0.1 + x[2] * (x[0] - 14.0);
    }
    __device__ static inline void sigma(const double *prm, const double (&)[D], const double *, double (&s)[D])
    {
        s[0] = prm[1]; s[1] = prm[1]; s[2] = prm[2];
    }
    __device__ static inline double stage(const double *, const double (&x)[D], const double *u)
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < D; i++) s = s + 1e2 * x[i] * x[i]; // the reference's order of operations (:139-141)
        return s + 1.0 * u[0] * u[0];
    }
    static constexpr bool STAGE_USEP = true;
    __device__ static inline double stage_x(const double *, const double (&x)[D])
    {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < D; i++) s = s + 1e2 * x[i] * x[i];
        return s;
    }
    __device__ static inline double stage_u(const double *, const double *u) { return 1.0 * u[0] * u[0]; }
    __device__ static inline double boundcost(const double *, const double (&)[D]) { return 1000.0; }
    __device__ static inline double obscost(const double *, const double (&)[D]) { return 0.0; }
};

// Universal model for arbitrary host callbacks (the reference's examples unchanged): the HOST evaluates the
// user's drift / diffusion / stage-cost callbacks for every (node, candidate) of the fibers it submits and the
// kernel reads the numbers from a table: per fiber [N][U][2D+1] = (drift[D], diag sigma[D], stage) and
// [N][2] = (boundcost, obscost).  c3sc_hip_bellman_fibers_tables.
template <int DIM>
struct TableModel {
    static constexpr int D = DIM, DU = 1;
    static constexpr int NTAB = 0, NCF = 0;
    static constexpr unsigned UDEP_MASK = 0, UCONST_MASK = 0;
    static constexpr bool IS_TABLE = true;
};

// FT stencil only (c3sc_hip_stencil_fibers): no dynamics
template <int DIM>
struct NoModel {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = DIM, DU = 1;
    static constexpr int NTAB = 0, NCF = 0;
    static constexpr unsigned UDEP_MASK = 0, UCONST_MASK = 0;
};

} // namespace c3sc
