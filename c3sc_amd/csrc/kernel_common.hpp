// kernel_common.hpp -- shared device code of the Bellman-backup kernels (gfx950, wave64).
// Citations are relative to the reference tree (goroda/c3sc).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/c3sc_hip.h"

namespace c3sc {

constexpr int MAXD = C3SC_MAX_DIM;

// Kernel argument block (passed by value; lives in SGPR/kernarg space, all fields wave-uniform).
struct KArgs {
    int d, k, N, ncand;
    long F;
    int ngrid[MAXD];
    int bctype[MAXD];
    // offsets (in doubles) into the read-only arena `ro` passed as a __restrict__ kernel
    // parameter, so that the compiler may keep wave-uniform reads on the scalar (SGPR) path
    int xg_off[MAXD];    // xgrid[m][0..N_m)
    long core_off[MAXD]; // rank-padded cores, see k_pad_core
    int nobs;
    int obs_off; // [nobs][2][d]: lb row then ub row per obstacle
    int cands_off; // [ncand][DU]
    int tab_off[4]; // model tables (univariate functions of a grid coordinate, host libm)
    int cfeat_off;  // [ncand][NCF] per-candidate features (host libm)
    double h2, discount;
    double t[2 * MAXD];
    double prm[C3SC_MAX_PARAMS];
    unsigned *status;
};

// Output pointers of one launch (separate __restrict__ kernel parameters).
struct KOut {
    double *out;       // [F][N]            (bellman kernel)
    double *costs;     // [F][N][2d+1]      (stencil kernel)
    int32_t *uidx;     // [F][N] or null
    int32_t *absorbed; // [F][N] or null
};

// Wave-level ordering of LDS traffic inside one wavefront (per-wave scratch, no s_barrier needed).
__device__ inline void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Neighbour indices of a FIXED dimension with index i on a grid of n nodes and boundary type bc:
// process_fibers_neighbor, nodeutil.c:513-566.  Returns true when the face is absorbing (the whole
// fiber is then absorbed, :520-522 / :542-544).  Periodic: node 0 and node n-1 are the same point (Q8).
__device__ inline bool fixed_neighbors(int i, int n, int bc, int &lo, int &hi)
{
    lo = i - 1;
    hi = i + 1;
    if (i == 0) {
        if (bc == C3SC_ABSORB) { lo = i; hi = i; return true; }
        if (bc == C3SC_REFLECT) { lo = i; hi = i + 1; }
        else { lo = n - 2; hi = i + 1; }
    } else if (i == n - 1) {
        if (bc == C3SC_ABSORB) { lo = i; hi = i; return true; }
        if (bc == C3SC_REFLECT) { lo = i - 1; hi = i; }
        else { lo = i - 1; hi = 1; }
    }
    return false;
}

// Neighbour indices of node j along the VARYING dimension and the final absorbed flag:
// nodeutil.c:570-624 (end points are overwritten by dim_vary's own boundary type -- quirk Q3).
// `ab_in` is the flag after the obstacle test and the fixed-face test.
__device__ inline int vary_neighbors(int j, int n, int bc, int ab_in, int &lo, int &hi)
{
    int ab = ab_in;
    if (j == 0) {
        if (bc == C3SC_ABSORB) { lo = 0; hi = 0; ab = 1; }
        else if (bc == C3SC_REFLECT) { lo = 0; hi = 1; ab = 0; }
        else { lo = n - 2; hi = 1; ab = 0; }
    } else if (j == n - 1) {
        if (bc == C3SC_ABSORB) { lo = n - 1; hi = n - 1; ab = 1; }
        else if (bc == C3SC_REFLECT) { lo = n - 2; hi = n - 1; ab = 0; }
        else { lo = n - 2; hi = 1; ab = 0; }
    } else if (ab == 0) {
        lo = j - 1;
        hi = j + 1;
    } else {
        lo = j;
        hi = j;
    }
    return ab;
}

// boundary_in_obstacle (boundary.c:668-680, bound_rect_inside :329-344): inclusive boxes.
template <int D>
__device__ inline bool in_obstacle(const KArgs &A, const double *__restrict__ ro, const double (&x)[D])
{
    int any = 0;
    for (int o = 0; o < A.nobs; o++) {
        const double *lb = ro + A.obs_off + (size_t)o * 2 * D;
        const double *ub = lb + D;
        int outside = 0; // branch-free: the reference's early exit (boundary.c:338-341) only saves time
#pragma unroll
        for (int m = 0; m < D; m++) outside |= (int)(x[m] < lb[m]) | (int)(x[m] > ub[m]);
        any |= (outside == 0);
    }
    return any != 0;
}

// The obstacle test split for kernels where one lane walks a fiber: bit o of the returned mask is set when
// the point is inside box o in every dim except `skip` (constant along the fiber) ...
template <int D>
__device__ inline unsigned obstacle_mask_fixed(const KArgs &A, const double *__restrict__ ro, const double (&x)[D], int skip)
{
    unsigned mask = 0;
    for (int o = 0; o < A.nobs; o++) {
        const double *lb = ro + A.obs_off + (size_t)o * 2 * D;
        const double *ub = lb + D;
        int outside = 0;
#pragma unroll
        for (int m = 0; m < D; m++)
            if (m != skip) outside |= (int)(x[m] < lb[m]) | (int)(x[m] > ub[m]);
        mask |= (outside == 0 ? 1u : 0u) << o;
    }
    return mask;
}
// ... and bit o set when coordinate xk of dim k lies inside box o's k-range (wave-uniform along the node loop)
template <int D>
__device__ inline unsigned obstacle_mask_dim(const KArgs &A, const double *__restrict__ ro, int k, double xk)
{
    unsigned mask = 0;
    for (int o = 0; o < A.nobs; o++) {
        const double lb = ro[A.obs_off + (size_t)o * 2 * D + k], ub = ro[A.obs_off + (size_t)o * 2 * D + D + k];
        mask |= ((xk < lb || xk > ub) ? 0u : 1u) << o;
    }
    return mask;
}

__device__ __forceinline__ void pin_vgpr(double &x) { asm volatile("" : "+v"(x)); }

// One node of the Bellman operator: bellman_optimal (bellman.c:504-543, BRUTEFORCE branch) over
// bellman_control (:367-480, no-gradient branch) = user dynamics + transition_assemble
// (nodeutil.c:267-406) + bellmanrhs (bellman.c:88-112).  V[2m], V[2m+1] = value at the (-,+)
// neighbour in dim m, V[2D] = value at the node.  The candidate scan keeps the first minimum
// (strict '<'), as the brute-force c3opt is assumed to.
template <class Model>
__device__ inline double node_backup(const KArgs &A, const double *__restrict__ ro, const double (&x)[Model::D],
                                     const int (&ix)[Model::D], const double (&V)[2 * Model::D + 1], int ab, int &ui,
                                     unsigned &st)
{
    constexpr int D = Model::D, DU = Model::DU;
    ui = -1;
    if (ab == 1) return Model::boundcost(A.prm, x);  // bellman.c:513-523
    if (ab == -1) return Model::obscost(A.prm, x);   // bellman.c:524-532
    constexpr int NT = Model::NTAB > 0 ? Model::NTAB : 1;
    const double *tab[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) tab[t] = ro + A.tab_off[t < Model::NTAB ? t : 0];
    typename Model::Node nd;
    Model::prep(A.prm, x, tab, ix, nd);
    // Everything that does not depend on the control is done once per node: the rates of the dims whose
    // drift / diffusion ignore u (Model::UDEP_MASK), their share of Q = sum p and of PV = sum p_i V_i
    // (nodeutil.c:289-309 fused with the ddot of bellman.c:95: rates are accumulated un-normalised; the
    // reference divides every p_i by Q first, nodeutil.c:397-402 -- same value up to a few ulp).
    constexpr unsigned UM = Model::UDEP_MASK;
    // scalars the candidate loop needs are parked in VGPRs: left as SGPRs the compiler spills whole
    // s_load tuples to VGPR lanes and restores them with dozens of v_readlane per candidate
    double tl[D], t2l[D], h2l = A.h2, discl = A.discount;
#pragma unroll
    for (int m = 0; m < D; m++) {
        tl[m] = A.t[2 * m];
        t2l[m] = A.t[2 * m + 1];
        if ((UM >> m) & 1u) { pin_vgpr(tl[m]); pin_vgpr(t2l[m]); }
    }
    pin_vgpr(h2l);
    pin_vgpr(discl);
    double Q0 = 0.0, PV0 = 0.0, stage0 = 0.0;
    {
        double u[DU], b[D], s[D];
#pragma unroll
        for (int i = 0; i < DU; i++) u[i] = ro[A.cands_off + i];
        Model::drift(A.prm, nd, x, u, ro + A.cfeat_off, b);
        Model::sigma(A.prm, x, u, s);
        if constexpr (!Model::STAGE_UDEP) stage0 = Model::stage(A.prm, x, u);
#pragma unroll
        for (int m = 0; m < D; m++) {
            if (!((UM >> m) & 1u)) {
                const double half = A.t[2 * m + 1] * (s[m] * s[m]) / 2.0;
                double pm = half, pp = half;
                if (b[m] < -1e-14) pm -= A.t[2 * m] * b[m];
                else if (b[m] > 1e-14) pp += A.t[2 * m] * b[m];
                Q0 += pm;
                Q0 += pp;
                PV0 = fma(pm, V[2 * m], PV0);
                PV0 = fma(pp, V[2 * m + 1], PV0);
            }
        }
    }
    double best = 0.0;
    for (int c = 0; c < A.ncand; c++) {
        double u[DU];
#pragma unroll
        for (int i = 0; i < DU; i++) u[i] = ro[A.cands_off + c * DU + i];
        double b[D], s[D];
        Model::drift(A.prm, nd, x, u, ro + A.cfeat_off + c * Model::NCF, b);
        Model::sigma(A.prm, x, u, s);
        const double stage = Model::STAGE_UDEP ? Model::stage(A.prm, x, u) : stage0;
        double Q = Q0, PV = PV0;
#pragma unroll
        for (int m = 0; m < D; m++) {
            if ((UM >> m) & 1u) {
                const double half = t2l[m] * (s[m] * s[m]) / 2.0;
                double pm = half, pp = half;
                if (b[m] < -1e-14) pm -= tl[m] * b[m];
                else if (b[m] > 1e-14) pp += tl[m] * b[m];
                Q += pm;
                Q += pp;
                PV = fma(pm, V[2 * m], PV);
                PV = fma(pp, V[2 * m + 1], PV);
            }
        }
        if (Q < 1e-14) { // nodeutil.c:365-367 returns 1; bellman.c:452 asserts.  Skip + flag.
            st |= C3SC_STATUS_STATIONARY;
            continue;
        }
        const double inv = 1.0 / Q;
        const double dt = h2l * inv;               // nodeutil.c:369
        const double pself = fma(-Q, inv, 1.0);    // 1 - sum_i p_i/Q: rounding residue, as in the reference
        const double ctg = fma(pself, V[2 * D], PV * inv);
        const double ebt = (A.discount == 0.0) ? 1.0 : exp(-discl * dt);      // bellman.c:94
        const double val = dt * stage + ebt * ctg;                            // bellman.c:97
        if (ui < 0 || val < best) {
            best = val;
            ui = c;
        }
    }
    return best;
}

} // namespace c3sc
