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
    // state dimensions each drift / diffusion entry reads (its tables count through their tab_dim), and those the costs read:
    // a rate whose inputs do not include the varying dimension is a constant of the fiber (kernel_fiber_pair.hpp: PairPark)
    static constexpr bool HAS_DEPS = true;
    __host__ __device__ static constexpr unsigned dep_mask(int m) { return m < 2 ? 1u << 2 : 0u; }
    static constexpr unsigned COST_DEP = 0u;
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
    // see Dubins3D: b0, b1 read v and theta; b2 omega; b3 a; b4 v, omega, delta; the stage cost x and y
    static constexpr bool HAS_DEPS = true;
    __host__ __device__ static constexpr unsigned dep_mask(int m)
    {
        return m < 2 ? (1u << 2) | (1u << 3) : (m == 2 ? 1u << 4 : (m == 3 ? 1u << 6 : (m == 4 ? (1u << 3) | (1u << 4) | (1u << 5) : 0u)));
    }
    static constexpr unsigned COST_DEP = 0x3u;
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
    // see Dubins3D: an even equation reads the next coordinate; the stage cost reads every coordinate
    static constexpr bool HAS_DEPS = true;
    __host__ __device__ static constexpr unsigned dep_mask(int m) { return (m % 2) == 0 ? 1u << ((m + 1 < DIM) ? m + 1 : m) : 0u; }
    static constexpr unsigned COST_DEP = (1u << DIM) - 1u;
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

// The 3-state / 3-control problem of the reference's own tests (test/transition_prob/tprob_test.c: drift f3 :223-251, diffusion
// s2 = I :197-220, stagecost3d :273-300, boundcost 100 :302-309, ocost 0 :311-318), used by Test_bellman_vi3d / Test_bellman_pi3d
// (:2361-2540) with u in [-5, 5]^3.  No params.
struct Tprob3D {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = 3, DU = 3;
    static constexpr int NTAB = 0, NCF = 0;
    static constexpr unsigned UDEP_MASK = 0x7u; // every equation carries a control
    static constexpr unsigned UCONST_MASK = 0;  // ... multiplied by the state
    static constexpr bool STAGE_UDEP = true;
    __host__ __device__ static constexpr int tab_dim(int) { return 0; }
    struct Node {};
    __device__ static inline void prep(const double *, const double (&)[D], const double (&)[1], Node &) {}
    __device__ static inline void drift(const double *, const Node &, const double (&x)[D], const double *u, const double *,
                                        double (&b)[D])
    {
        b[0] = x[0] * (x[2] * x[2]) * u[0];     // x[0]*pow(x[2],2)*u[0]
        b[1] = -x[1] * u[2] + u[1];
        b[2] = x[0] * x[1] * u[0] + 2 * u[1];
    }
    __device__ static inline void sigma(const double *, const double (&)[D], const double *, double (&s)[D])
    {
        s[0] = 1.0; s[1] = 1.0; s[2] = 1.0;
    }
    __device__ static inline double stage(const double *, const double (&x)[D], const double *u)
    { // the reference's order of accumulation (:283-291)
        double s = 0.0;
        s += 0.2 * x[0] * x[0];
        s += 0.5 * x[1] * x[1];
        s += 2.0 * x[2] * x[2];
        s += 0.1 * u[0] * u[0];
        s += 0.5 * u[1] * u[1];
        s += 3.0 * u[2] * u[2];
        return s;
    }
    __device__ static inline double boundcost(const double *, const double (&)[D]) { return 100.0; }
    __device__ static inline double obscost(const double *, const double (&)[D]) { return 0.0; }
};

// examples/perching/perch.c:36-273 -- the reference's one real 7-D problem: a glider perching.  State (x, z, theta, phi, dx, dz,
// dtheta), control = elevator rate.  The reference computes the aerodynamic forces as rho S |v|^2 sin(alpha) with
// alpha = angle - atan2(v_z, v_x) (:104-109); sin(a - atan2(y, x)) = (sin a x - cos a y) / |v|, so the force is
// rho S |v| (sin a v_x - cos a v_z): a square root instead of atan2 + sin per candidate, equal to rounding (the oracle keeps the
// reference's libm form; parity holds to 1e-12 of the value scale).  Tables: cos / sin of theta (dim 2) and of phi (dim 3);
// cos / sin of theta + phi by the addition formulas.  Diffusion 1e-9 I (:165-176); no params.
struct Perch7D {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = 7, DU = 1;
    static constexpr int NTAB = 4, NCF = 0;
    static constexpr unsigned UDEP_MASK = (1u << 3) | (1u << 4) | (1u << 5) | (1u << 6);
    static constexpr unsigned UCONST_MASK = 1u << 3; // dphi/dt = u alone; the accelerations also depend on the state
    static constexpr bool STAGE_UDEP = true;
    __host__ __device__ static constexpr int tab_dim(int t) { return t < 2 ? 2 : 3; }
    struct Node { double ct, st, cp, ctp, stp, fw, ex0, ex1; };
    __device__ static inline void prep(const double *, const double (&x)[D], const double (&tv)[4], Node &n)
    {
        const double l = 0.35, l_w = -0.03, rho = 1.292, S_w = 0.1;
        n.ct = tv[0]; n.st = tv[1]; n.cp = tv[2];
        const double sp = tv[3];
        n.ctp = n.ct * n.cp - n.st * sp;
        n.stp = n.st * n.cp + n.ct * sp;
        const double w0 = x[4] + l_w * x[6] * n.st, w1 = x[5] - l_w * x[6] * n.ct; // wing velocity (:93-95)
        n.fw = rho * S_w * sqrt(w0 * w0 + w1 * w1) * (n.st * w0 - n.ct * w1);
        n.ex0 = x[4] + l * x[6] * n.st;                                              // elevator velocity without the control (:100-102)
        n.ex1 = x[5] - l * x[6] * n.ct;
    }
    __device__ static inline void drift(const double *, const Node &n, const double (&x)[D], const double *u, const double *,
                                        double (&b)[D])
    {
        const double m = 0.05, g = 9.81, rho = 1.292, S_e = 0.025, In = 6e-3, l = 0.35, l_w = -0.03, l_e = 0.04;
        const double e0 = n.ex0 + l_e * (x[6] + u[0]) * n.stp, e1 = n.ex1 - l_e * (x[6] + u[0]) * n.ctp;
        const double fe = rho * S_e * sqrt(e0 * e0 + e1 * e1) * (n.stp * e0 - n.ctp * e1);
        b[0] = x[4]; b[1] = x[5]; b[2] = x[6]; b[3] = u[0];
        b[4] = (-n.fw * n.st - fe * n.stp) / m;
        b[5] = (n.fw * n.ct + fe * n.ctp - m * g) / m;
        b[6] = (-n.fw * l_w - fe * (l * n.cp + l_e)) / In;
    }
    __device__ static inline void sigma(const double *, const double (&)[D], const double *, double (&s)[D])
    {
#pragma unroll
        for (int i = 0; i < D; i++) s[i] = 1e-9;
    }
    __device__ static inline double stage(const double *, const double (&x)[D], const double *u)
    { // :185-215, in the reference's order
        double s = 0.0;
        s += 20.0 * x[0] * x[0];
        s += 50.0 * x[1] * x[1];
        s += 10.0 * x[2] * x[2];
        s += 1.0 * x[3] * x[3];
        s += 1.0 * x[4] * x[4];
        s += 1.0 * x[5] * x[5];
        s += 1.0 * x[6] * x[6];
        s += 0.1 * u[0] * u[0];
        return s;
    }
    __device__ static inline double boundcost(const double *, const double (&x)[D])
    { // :222-241
        const double hp = 1.5707963267948966; // M_PI / 2.0
        double s = 0.0;
        s += 600.0 * x[0] * x[0];
        s += 400.0 * x[1] * x[1];
        s += 1.0 / 9.0 * x[2] * x[2];
        s += 5.0 * (x[2] - hp) * (x[2] - hp);
        s += 1.0 / 9.0 * x[3] * x[3];
        s += 1.0 * x[4] * x[4];
        s += 1.0 * (x[5] + 1.5) * (x[5] + 1.5);
        s += 1.0 / 9.0 * (x[6] + 0.5) * (x[6] + 0.5);
        return s;
    }
    __device__ static inline double obscost(const double *, const double (&)[D]) { return 0.0; }
};

// examples/skidding5d/scar.c:39-176 (order = {0,1,2,3,4}): the skidding car with lateral dynamics -- state (x, y, orientation,
// yaw rate, lateral speed) at the constant forward speed s = 27, control = steering angle.  Diffusion (:111-134): the callback
// zero-fills its 5 x 5 matrix and then writes out[0], out[6], out[12], out[28] (sic: outside the matrix) and
// out[24] -- the diagonal entry out[18] of the yaw rate is never set, so its noise is 0; mirrored here, not fixed.
struct Skid5D {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = 5, DU = 1;
    static constexpr int NTAB = 2, NCF = 0; // tables: cos(x2), sin(x2)
    static constexpr unsigned UDEP_MASK = (1u << 3) | (1u << 4); // the front tyre force carries the steering angle
    static constexpr unsigned UCONST_MASK = 0;                   // ... next to the state
    static constexpr bool STAGE_UDEP = false;
    __host__ __device__ static constexpr int tab_dim(int) { return 2; }
    struct Node { double b0, b1, fs, ft; };
    __device__ static inline void prep(const double *, const double (&x)[D], const double (&tv)[2], Node &n)
    {
        const double cf = 17000.0, ct = 20000.0, a = 1.2, b = 1.5, s = 27.0;
        const double angvel = x[3], speed = x[4];
        n.b0 = s * tv[0] - speed * tv[1];          // s * co - speed * so (:82)
        n.b1 = s * tv[1] + speed * tv[0];          // s * so + speed * co (:83)
        n.fs = (speed + a * angvel) / s;           // the state's share of ff = cf * ((speed + a * angvel) / s + steering) (:76)
        n.ft = ct * (speed - b * angvel) / s;      // (:77)
        (void)cf;
    }
    __device__ static inline void drift(const double *, const Node &n, const double (&x)[D], const double *u, const double *,
                                        double (&bb)[D])
    {
        const double m = 1460.0, cf = 17000.0, a = 1.2, b = 1.5, In = 2170.0, s = 27.0;
        const double ff = cf * (n.fs + u[0]);
        bb[0] = n.b0; bb[1] = n.b1; bb[2] = x[3];
        bb[3] = (a * ff - b * n.ft) / In;          // (:85)
        bb[4] = -s * x[3] + (ff + n.ft) / m;       // (:86)
    }
    __device__ static inline void sigma(const double *, const double (&)[D], const double *, double (&sg)[D])
    {
        sg[0] = 1e-5; sg[1] = 1e-5; sg[2] = 1e-5; sg[3] = 0.0; sg[4] = 1e-5; // Q13: out[28] instead of out[18]
    }
    __device__ static inline double stage(const double *, const double (&x)[D], const double *)
    { // :143-144, the reference's order
        double o = 1.0 + 0.02 * (x[0] * x[0]) + 0.02 * (x[1] * x[1]);
        o = o + x[3] * x[3] + x[4] * x[4];
        return o;
    }
    __device__ static inline double boundcost(const double *, const double (&x)[D])
    { // :158-159
        double o = 0.1 * (x[0] * x[0]) + 0.1 * (x[1] * x[1]);
        o = o + 0.1 * (x[3] * x[3]) + 0.1 * (x[4] * x[4]);
        return o;
    }
    __device__ static inline double obscost(const double *, const double (&)[D]) { return 0.0; }
};

// examples/cothrust2/copterposethrust.c:40-222 (order = {0,..,5}): quadcopter position (x, y, z) and velocity, controls (thrust,
// roll phi, pitch theta).  The three accelerations are functions of the control alone (:115-117): they are the candidate's FEATURES,
// evaluated on the host with libm exactly as the callback writes them (cos / sin of the angles included), so candidate lists need
// no transcendental on the device; the continuous-control (box) minimiser evaluates the same expressions with the device's
// sin / cos (parity to the minimiser's tolerance, as for every box run).  Diffusion diag (:147-152), stage (:171-176).
struct Cothrust6D {
    static constexpr bool IS_TABLE = false;
    static constexpr int D = 6, DU = 3;
    static constexpr int NTAB = 0, NCF = 3; // features: the accelerations b3, b4, b5 of the candidate
    static constexpr unsigned UDEP_MASK = (1u << 3) | (1u << 4) | (1u << 5);
    static constexpr unsigned UCONST_MASK = UDEP_MASK; // ... of the control alone: rates are per-candidate constants
    static constexpr bool STAGE_UDEP = true;
    static constexpr bool CF_FROM_U = true; // the features can be formed from u on the device (box minimiser)
    __host__ __device__ static constexpr int tab_dim(int) { return 0; }
    struct Node {};
    __device__ static inline void prep(const double *, const double (&)[D], const double (&)[1], Node &) {}
    __device__ static inline void features(const double *u, double *cf)
    {
        const double m = 1.227, g = 9.81, mg = m * g;
        const double cphi = cos(u[1]), sphi = sin(u[1]), cth = cos(u[2]), sth = sin(u[2]);
        cf[0] = cphi * sth * (u[0] - mg) / m;
        cf[1] = -sphi * (u[0] - mg) / m;
        cf[2] = g + cth * cphi * (u[0] - mg) / m;
    }
    __device__ static inline void drift(const double *, const Node &, const double (&x)[D], const double *, const double *cf,
                                        double (&b)[D])
    {
        b[0] = x[3]; b[1] = x[4]; b[2] = x[5]; b[3] = cf[0]; b[4] = cf[1]; b[5] = cf[2];
    }
    __device__ static inline void sigma(const double *, const double (&)[D], const double *, double (&s)[D])
    {
        s[0] = 1e-1; s[1] = 1e-1; s[2] = 2e-1; s[3] = 12e-1; s[4] = 12e-1; s[5] = 12e-1;
    }
    __device__ static inline double stage(const double *, const double (&x)[D], const double *u)
    { // :171-176, the reference's order
        double o = 0.0;
        o = o + 60.0 + 2.0 * (u[0] * u[0]) + 1.0 * (u[1] * u[1]) + 6.0 * (u[2] * u[2]);
        o = o + 8.0 * (x[2] * x[2]);
        o = o + 6.0 * (x[1] * x[1]);
        o = o + 8.0 * (x[0] * x[0]);
        return o;
    }
    __device__ static inline double boundcost(const double *, const double (&)[D]) { return 10.0; }
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
