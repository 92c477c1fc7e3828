// kernel_common.hpp -- shared device code of the Bellman-backup kernels (gfx950, wave64).
// Citations are relative to the reference tree (goroda/c3sc).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/c3sc_hip.h"

// in-kernel cycle stamps / ablation switches exist only in diagnostic builds (make STAMPS=1)
#ifdef C3SC_STAMPS
#define C3SC_STAMPS_ON 1
#else
#define C3SC_STAMPS_ON 0
#endif

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
    unsigned long long *dbgbuf; // [waves][8] segment cycle sums when (dbg & 128)
    const int32_t *forced;      // policy evaluation (bellman_pi): [F][N] candidate index to apply per node, or null = minimise
    int dbg; // ablation switches for profiling builds (0 in production): see kernel_fiber_pair.hpp
    // continuous controls in a box (bellman_optimal's non-BRUTEFORCE branch, bellman.c:545-1118): cmode = 1
    int cmode, ugrid, upolish; // grid points per control dim, polish rounds
    double ulb[C3SC_MAX_DU], uub[C3SC_MAX_DU];
    double *uopt;              // [F][N][du] minimiser per node (may be null)
    const double *forced_u;    // policy evaluation with continuous controls: [F][N][du] control to apply, or null
    int tbl_off;               // fiber-pair kernel: offset (doubles) of the candidate / node tables in dynamic LDS
    // fiber-quad kernel (kernel_fiber_quad.hpp): extra copies of the cores made by k_quad_aux, offsets into `ro`
    long quad_coreT_off[MAXD]; // middle cores row-major (a*RP + b): the staged matrix of the suffix-side levels
    long quad_aop_off[MAXD];   // middle cores as MFMA A operands: [N][c | a][MB][C][64]
    int quad_sv_off;           // offset (doubles) of the per-wave node-value rows in dynamic LDS
    int quad_x_off;            // duo kernel: offset (doubles) of the pairs' exchange rows [2D][64] in dynamic LDS
    int quad_m1_off;           // duo kernel: offset (doubles) of the second staging buffer; quad_ix_off: the pairs' fiber indices
    int quad_ix_off;
    long quad_imgL_off[MAXD];  // duo kernel: LDS images (node stride elems + 2) of the cores as the prefix side stages them
    long quad_imgR_off[MAXD];  //             and of the transposed cores (suffix side); read through img_base by LDS-DMA
    long pair_img_off[MAXD];   // every core once more as the fiber-pair kernel's LDS image: [N][elems | 1] (k_core_image)
    int cends;                 // c3sc_hip_set_consistent_ends: see vary_neighbors
    // node memo of the device-resident cross iterations (cross_device.hip), applied in the epilogue of the fiber-per-wave kernel
    // when memo_keys != null: a node already stored in this sweep takes the stored value, a new one is stored
    // (bellman.c:1333-1353, 1412-1417)
    unsigned long long *memo_keys;
    double *memo_vals;
    unsigned long long memo_capmask, memo_epoch_bits;
    int memo_shift;
    long long memo_stride[MAXD];
    unsigned long long *memo_counters; // [0] nodes stored, [3] table full
    const int *skip; // fiber-per-wave kernel: return at once when *skip != 0 (the caller already holds this batch's values)
    int memo_mode; // 0: the node's VALUE (bellman_vi's memo); 1: its POLICY, the winning candidate index of a live node
                   // (bellman_pi's per-node cache under key2, bellman.c:1806, 1877)
    const double *img_base;    // the arena again, as a pointer that is NOT the kernels' `ro` argument: the LDS-DMA copy reads the
                               // images through it (see stage_core_image on why it must not be derived from `ro`)
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
    // selects only: i differs per lane and the kernels that keep wave-uniform tables in VGPR lanes
    // (NodeRegs / CandRegs) must not run lane-divergent branches (see node_backup)
    const bool first = (i == 0), last = (i == n - 1) & !first;
    const bool absorb = (bc == C3SC_ABSORB), reflect = (bc == C3SC_REFLECT);
    const int lo_first = absorb ? i : (reflect ? i : n - 2);
    const int hi_first = absorb ? i : i + 1;
    const int lo_last = absorb ? i : i - 1;
    const int hi_last = absorb ? i : (reflect ? i : 1);
    lo = first ? lo_first : (last ? lo_last : i - 1);
    hi = first ? hi_first : (last ? hi_last : i + 1);
    return (first | last) & absorb;
}

// Neighbour indices of node j along the VARYING dimension and the final absorbed flag:
// nodeutil.c:570-624 (end points are overwritten by dim_vary's own boundary type -- quirk Q3).
// `ab_in` is the flag after the obstacle test and the fixed-face test.
// keep_ends (KArgs::cends, c3sc_hip_set_consistent_ends; NOT the reference's behaviour, default 0): the end points of a
// reflecting / periodic fiber keep `ab_in` instead of being reset to 0, i.e. the flag every other fiber direction gives that
// node -- the value of a node then no longer depends on the direction of the fiber it is computed in.
__device__ inline int vary_neighbors(int j, int n, int bc, int ab_in, int &lo, int &hi, int keep_ends = 0)
{
    int ab = ab_in;
    const int ab_end = keep_ends ? ab_in : 0;
    if (j == 0) {
        if (bc == C3SC_ABSORB) { lo = 0; hi = 0; ab = 1; }
        else if (bc == C3SC_REFLECT) { lo = 0; hi = 1; ab = ab_end; }
        else { lo = n - 2; hi = 1; ab = ab_end; }
    } else if (j == n - 1) {
        if (bc == C3SC_ABSORB) { lo = n - 1; hi = n - 1; ab = 1; }
        else if (bc == C3SC_REFLECT) { lo = n - 2; hi = n - 1; ab = ab_end; }
        else { lo = n - 2; hi = 1; ab = ab_end; }
    } else if (ab == 0) {
        lo = j - 1;
        hi = j + 1;
    } else {
        lo = j;
        hi = j;
    }
    return ab;
}

// Node memo (open addressing, linear probing).  Key word: [63:49] epoch of the sweep (never 0) | [48] pending | [47:0] node id;
// a slot whose epoch is not the current one is free, so the table is cleared by advancing the epoch.  A hit returns the stored
// value (first value stays, bellman.c:1349-1353), a miss stores v.  The same node twice in one batch is the same fiber twice
// (a batch varies one dimension), i.e. identical values: the loser of that race keeps its own.  *inserted reports a store.
constexpr unsigned long long MEMO_PENDING = 1ull << 48, MEMO_EPOCH_MASK = ~((1ull << 49) - 1), MEMO_ID_MASK = (1ull << 48) - 1;
__device__ inline double memo_merge(unsigned long long *keys, double *vals, unsigned long long capmask, int shift, unsigned long long epoch_bits,
                                    unsigned long long id, double v, int &inserted, int &overflow)
{
    const unsigned long long K = epoch_bits | id, KP = K | MEMO_PENDING;
    unsigned long long slot = (id * 0x9E3779B97F4A7C15ull) >> shift;
    for (unsigned long long probe = 0; probe <= capmask; probe++, slot = (slot + 1) & capmask) {
        unsigned long long cur = __hip_atomic_load(&keys[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        while ((cur & MEMO_EPOCH_MASK) != epoch_bits) { // free: claim it
            const unsigned long long prev = atomicCAS(&keys[slot], cur, KP);
            if (prev == cur) {
                vals[slot] = v;
                __hip_atomic_store(&keys[slot], K, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                inserted = 1;
                return v;
            }
            cur = prev;
        }
        if ((cur & ~MEMO_PENDING) == K) // this node: stored by an earlier batch (complete) or by a twin of this batch (pending)
            return (cur & MEMO_PENDING) ? v : __hip_atomic_load(&vals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    overflow = 1;
    return v;
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

// Wave-uniform tables without memory latency: lane l keeps entry l in a VGPR, v_readlane_b32 broadcasts the
// entry selected by a wave-uniform index into SGPRs (two 32-bit halves per double).
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Lane-distributed per-node data of the varying dimension K (N_K <= 128): coordinate, obstacle k-range
// mask and the model tables indexed by dim K.  Everything the node loop needs per node is then a
// v_readlane away instead of a scalar-memory round trip.
template <class Model, int K>
struct NodeRegs {
    static constexpr int NT = Model::NTAB > 0 ? Model::NTAB : 1;
    __host__ __device__ static constexpr int nkt()
    { // number of tables indexed by dim K
        int n = 0;
        for (int t = 0; t < Model::NTAB; t++) n += (Model::tab_dim(t) == K);
        return n;
    }
    __host__ __device__ static constexpr int kslot(int t)
    { // slot of table t among the K-indexed tables
        int n = 0;
        for (int q = 0; q < t; q++) n += (Model::tab_dim(q) == K);
        return n;
    }
    static constexpr int NKT = nkt() > 0 ? nkt() : 1;
    double xk[2];
    unsigned km[2];
    double tk[NKT][2];
    __device__ inline void load(const KArgs &A, const double *__restrict__ ro)
    {
        constexpr int D = Model::D;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int j = min((int)(threadIdx.x & 63) + 64 * q, A.N - 1);
            xk[q] = ro[A.xg_off[K] + j];
            km[q] = obstacle_mask_dim<D>(A, ro, K, xk[q]);
#pragma unroll
            for (int t = 0; t < Model::NTAB; t++)
                if (Model::tab_dim(t) == K) tk[kslot(t)][q] = ro[A.tab_off[t] + j];
        }
    }
    __device__ inline double x_at(int j) const { return j < 64 ? readlane_f64(xk[0], j) : readlane_f64(xk[1], j - 64); }
    __device__ inline unsigned mask_at(int j) const
    {
        return j < 64 ? (unsigned)__builtin_amdgcn_readlane((int)km[0], j) : (unsigned)__builtin_amdgcn_readlane((int)km[1], j - 64);
    }
    __device__ inline double tab_at(int t, int j) const
    {
        return j < 64 ? readlane_f64(tk[kslot(t)][0], j) : readlane_f64(tk[kslot(t)][1], j - 64);
    }
};

// Lane-distributed copy of the control candidates and their features (lane c holds candidate c0 + c: 64 candidates at a time;
// the fiber-per-wave kernel walks longer lists in chunks, the other kernels decline them),
// plus the upwind rates of the dims whose drift and diffusion depend on the control alone (Model::UCONST_MASK):
// those are constants of the candidate (nodeutil.c:289-309 with b = b(u), sigma = sigma(u)), computed once per
// wave with the very arithmetic the node loop would use and broadcast by v_readlane in the scan.
// models whose stage cost splits as stage_x(x) + stage_u(u) (and have no control features, NCF == 0) keep stage_u in the
// otherwise unused feature slot of the candidate table
template <class Model>
constexpr bool stage_usep()
{
    if constexpr (requires { Model::STAGE_USEP; }) return Model::STAGE_USEP && Model::NCF == 0;
    else return false;
}

template <class Model>
struct CandRegs {
    static constexpr unsigned UC = Model::UCONST_MASK;
    __host__ __device__ static constexpr int nuc()
    {
        int n = 0;
        for (int m = 0; m < Model::D; m++) n += (UC >> m) & 1u;
        return n;
    }
    __host__ __device__ static constexpr int ucslot(int m)
    {
        int n = 0;
        for (int q = 0; q < m; q++) n += (UC >> q) & 1u;
        return n;
    }
    static constexpr int NUC = nuc() > 0 ? nuc() : 1;
    double u[Model::DU];
    double cf[Model::NCF > 0 ? Model::NCF : 1];
    double rpm[NUC], rpp[NUC]; // rate to the (-,+) neighbour of each UCONST dim
    double qab;                // their sum, in the association the scan uses: (sum pm) + (sum pp)
    // accessors shared with CandLds (the same table in LDS): candidate c is wave-uniform
    __device__ inline double get_u(int i, int c) const { return readlane_f64(u[i], c); }
    __device__ inline double get_cf(int i, int c) const { return readlane_f64(cf[i], c); }
    __device__ inline double get_rpm(int slot, int c) const { return readlane_f64(rpm[slot], c); }
    __device__ inline double get_rpp(int slot, int c) const { return readlane_f64(rpp[slot], c); }
    __device__ inline double get_qab(int c) const { return readlane_f64(qab, c); }
    __device__ inline void load(const KArgs &A, const double *__restrict__ ro, int c0 = 0) // candidates c0 .. c0 + 63
    {
        constexpr int D = Model::D;
        const int c = min(c0 + (int)(threadIdx.x & 63), A.ncand - 1);
#pragma unroll
        for (int i = 0; i < Model::DU; i++) u[i] = ro[A.cands_off + c * Model::DU + i];
        cf[0] = 0.0;
        if constexpr (stage_usep<Model>()) cf[0] = Model::stage_u(A.prm, u);
#pragma unroll
        for (int i = 0; i < Model::NCF; i++) cf[i] = ro[A.cfeat_off + c * Model::NCF + i];
        rpm[0] = rpp[0] = qab = 0.0;
        if constexpr (UC != 0) {
            typename Model::Node nd{};
            double x[D], b[D], s[D];
#pragma unroll
            for (int m = 0; m < D; m++) x[m] = 0.0;
            Model::drift(A.prm, nd, x, u, cf, b); // only the UCONST entries are meaningful and used
            Model::sigma(A.prm, x, u, s);
            double Qa = 0.0, Qb = 0.0;
#pragma unroll
            for (int m = 0; m < D; m++) {
                if ((UC >> m) & 1u) {
                    const double half = A.t[2 * m + 1] * (s[m] * s[m]) / 2.0;
                    const double tb = A.t[2 * m] * b[m];
                    const double pm = (b[m] < -1e-14) ? half - tb : half;
                    const double pp = (b[m] > 1e-14) ? half + tb : half;
                    rpm[ucslot(m)] = pm;
                    rpp[ucslot(m)] = pp;
                    Qa += pm;
                    Qb += pp;
                }
            }
            qab = Qa + Qb;
        }
    }
};

// The same candidate table in LDS (one row of CW doubles per candidate), read with a wave-uniform address (broadcast):
// no VGPRs held across the node loop and nothing that depends on inactive lanes.  Filled once per workgroup from a
// CandRegs (every lane writes its own candidate's row; lanes past ncand repeat the last candidate).
template <class Model>
struct CandLds {
    static constexpr int DU = Model::DU, NCFa = Model::NCF > 0 ? Model::NCF : 1, NUC = CandRegs<Model>::NUC;
    static constexpr int CW = DU + NCFa + 2 * NUC + 1;
    const double *tb;
    __host__ __device__ static constexpr int doubles(int ncand) { return ncand * CW; }
    __device__ inline void fill(double *dst, const CandRegs<Model> &cr, int ncand, int c0 = 0) // cr holds candidates c0 .. c0 + 63
    {
        const int c = min(c0 + (int)(threadIdx.x & 63), ncand - 1);
        double *row = dst + c * CW;
#pragma unroll
        for (int i = 0; i < DU; i++) row[i] = cr.u[i];
#pragma unroll
        for (int i = 0; i < NCFa; i++) row[DU + i] = cr.cf[i];
#pragma unroll
        for (int i = 0; i < NUC; i++) { row[DU + NCFa + i] = cr.rpm[i]; row[DU + NCFa + NUC + i] = cr.rpp[i]; }
        row[DU + NCFa + 2 * NUC] = cr.qab;
        tb = dst;
    }
    __device__ inline double get_u(int i, int c) const { return tb[c * CW + i]; }
    __device__ inline double get_cf(int i, int c) const { return tb[c * CW + DU + i]; }
    __device__ inline double get_rpm(int slot, int c) const { return tb[c * CW + DU + NCFa + slot]; }
    __device__ inline double get_rpp(int slot, int c) const { return tb[c * CW + DU + NCFa + NUC + slot]; }
    __device__ inline double get_qab(int c) const { return tb[c * CW + DU + NCFa + 2 * NUC]; }
};

// NodeRegs in LDS: per node of the varying dim (x, K-indexed tables..., obstacle mask as a double-sized slot)
template <class Model, int K>
struct NodeLds {
    static constexpr int NKT = NodeRegs<Model, K>::NKT;
    static constexpr int NW = 2 + NKT; // x, mask, tables
    const double *tb;
    __host__ __device__ static constexpr int doubles(int N) { return N * NW; }
    __device__ inline void fill(double *dst, const NodeRegs<Model, K> &nr, int N)
    {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int j = min((int)(threadIdx.x & 63) + 64 * q, N - 1);
            double *row = dst + j * NW;
            row[0] = nr.xk[q];
            row[1] = __hiloint2double(0, (int)nr.km[q]);
#pragma unroll
            for (int t = 0; t < NKT; t++) row[2 + t] = nr.tk[t][q];
        }
        tb = dst;
    }
    __device__ inline double x_at(int j) const { return tb[j * NW]; }
    __device__ inline unsigned mask_at(int j) const { return (unsigned)__double2loint(tb[j * NW + 1]); }
    __device__ inline double tab_at(int t, int j) const { return tb[j * NW + 2 + NodeRegs<Model, K>::kslot(t)]; }
};

// values of the model's tables at the node with grid indices ix[]
template <class Model>
__device__ inline void table_values(const KArgs &A, const double *__restrict__ ro, const int (&ix)[Model::D],
                                    double (&tv)[Model::NTAB > 0 ? Model::NTAB : 1])
{
    tv[0] = 0.0;
#pragma unroll
    for (int t = 0; t < Model::NTAB; t++) tv[t] = ro[A.tab_off[t] + ix[Model::tab_dim(t)]];
}

// exp(x) for the discount factor exp(-beta dt) (bellman.c:94).  With an MCA time step dt = h^2/Q the argument is tiny
// (|x| ~ 1e-4 for the LQG examples); below 2^-7 a degree-7 Taylor polynomial is exact to double rounding (the first
// omitted term is < 2^-56/40320 relative), so the ~30-instruction libm path runs only when some lane needs it
// (wave-uniform branch).
__device__ __forceinline__ double sgpr_const(double c)
{ // materialised on the scalar unit where it is used: a literal the compiler hoists out of the node loop sits in a
  // VGPR pair for the whole kernel (seven of them cost the undiscounted car7d kernel 11 %)
    asm volatile("" : "+s"(c));
    return c;
}

// exp(x) for |x| < 2^-7: degree-7 Taylor polynomial (truncation 2^-56 * 1/40320 relative, below an ulp)
__device__ inline double exp_small(double x)
{
    double p = sgpr_const(1.0 / 5040.0);
    p = fma(p, x, sgpr_const(1.0 / 720.0));
    p = fma(p, x, sgpr_const(1.0 / 120.0));
    p = fma(p, x, sgpr_const(1.0 / 24.0));
    p = fma(p, x, sgpr_const(1.0 / 6.0));
    p = fma(p, x, 0.5);
    p = fma(p, x, 1.0);
    return fma(p, x, 1.0);
}

// exp(x) for |x| < 2^-10: degree 4 (truncation x^5/120 < 2^-56, below an ulp)
__device__ inline double exp_tiny(double x)
{
    double p = sgpr_const(1.0 / 24.0);
    p = fma(p, x, sgpr_const(1.0 / 6.0));
    p = fma(p, x, 0.5);
    p = fma(p, x, 1.0);
    return fma(p, x, 1.0);
}

__device__ inline double exp_discount(double x)
{
    if (__all(fabs(x) < 0.0078125)) return exp_small(x);
    return exp(x);
}

// One node of the Bellman operator from host-evaluated callback tables (TableModel): row = this node's
// [U][2D+1] block, cost = (boundcost, obscost).  Same arithmetic as node_backup below.
template <int D>
__device__ inline double node_backup_tables(const KArgs &A, const double *__restrict__ row, const double *__restrict__ cost,
                                            const double (&V)[2 * D + 1], int ab, int &ui, unsigned &st, bool forced = false,
                                            int fu = -1)
{
    ui = -1;
    if (ab == 1) return cost[0];
    if (ab == -1) return cost[1];
    double best = 0.0;
    for (int c = 0; c < A.ncand; c++) {
        const double *r = row + (size_t)c * (2 * D + 1);
        double Q = 0.0, PV = 0.0;
#pragma unroll
        for (int m = 0; m < D; m++) {
            const double b = r[m], s = r[D + m];
            const double half = A.t[2 * m + 1] * (s * s) / 2.0;
            const double tb = A.t[2 * m] * b;
            const double pm = (b < -1e-14) ? half - tb : half;
            const double pp = (b > 1e-14) ? half + tb : half;
            Q += pm;
            Q += pp;
            PV = fma(pm, V[2 * m], PV);
            PV = fma(pp, V[2 * m + 1], PV);
        }
        if (Q < 1e-14) { st |= C3SC_STATUS_STATIONARY; continue; }
        const double inv = 1.0 / Q;
        const double dt = A.h2 * inv;
        const double pself = fma(-Q, inv, 1.0);
        const double ctg = fma(pself, V[2 * D], PV * inv);
        const double ebt = (A.discount == 0.0) ? 1.0 : exp(-A.discount * dt);
        const double val = dt * r[2 * D] + ebt * ctg;
        if (forced ? (c == fu) : (ui < 0 || val < best)) { best = val; ui = c; }
    }
    return best;
}

// 1/q for a positive, normal q: hardware seed + two Newton steps (5 instructions; the IEEE division sequence with its
// scaling and fix-up is 12, and the discounted scan divides once per candidate)
__device__ inline double rcp_newton(double q)
{
    double x = __builtin_amdgcn_rcp(q);
    x = fma(fma(-q, x, 1.0), x, x);
    x = fma(fma(-q, x, 1.0), x, x);
    return x;
}

// One node of the Bellman operator: bellman_optimal (bellman.c:504-543, BRUTEFORCE branch) over
// bellman_control (:367-480, no-gradient branch) = user dynamics + transition_assemble
// (nodeutil.c:267-406) + bellmanrhs (bellman.c:88-112).  V[2m], V[2m+1] = value at the (-,+)
// neighbour in dim m, V[2D] = value at the node.  The candidate scan keeps the first minimum
// (strict '<'), as the brute-force c3opt is assumed to.
// precomputed rates of control-independent dimensions (NoPre: none).  MASK: the dimensions whose (p-, p+) come from pm(m) / pp(m)
// instead of being formed from the node's drift -- the fiber-pair kernel keeps those that do not depend on the varying
// dimension per fiber (PairPark)
struct NoPre {
    static constexpr unsigned MASK = 0u;
    __device__ inline double pm(int) const { return 0.0; }
    __device__ inline double pp(int) const { return 0.0; }
};

// upwind rates of one dimension (nodeutil.c:289-309): p-+ = t2 sigma^2 / 2 + t max(-+b, 0) with the +-1e-14 dead zone
__device__ __forceinline__ void upwind_rates(double t, double t2, double b, double sg, double &pm, double &pp)
{
    const double half = t2 * (sg * sg) / 2.0;
    const double tb = t * b;
    pm = (b < -1e-14) ? half - tb : half;
    pp = (b > 1e-14) ? half + tb : half;
}

template <class Model, int CG = 1, int CGD = 1, class Cand = CandRegs<Model>, bool SPLIT = true, class Pre = NoPre>
__device__ inline double node_backup(const KArgs &A, const double *__restrict__ ro, const double (&x)[Model::D],
                                     const double (&tv)[Model::NTAB > 0 ? Model::NTAB : 1], const Cand &cr,
                                     const double (&V)[2 * Model::D + 1], int ab, int &ui, unsigned &st, bool forced = false,
                                     int fu = -1, const Pre &pre = Pre())
{
    // forced (wave-uniform) = policy evaluation, bellman_pi (bellman.c:1702-1886): the candidate fu (per lane) is
    // applied instead of the minimiser's; same rates, same bellmanrhs.
    constexpr int D = Model::D, DU = Model::DU;
    const int nc = __builtin_amdgcn_readfirstlane(A.ncand); // keeps the candidate loops' trip test on the scalar unit
    ui = -1;
    // Absorbed lanes (bellman.c:513-532) do not leave early: the scan below runs with the whole wave active and
    // the absorbed lanes' result is replaced at the end.  The lane-distributed tables (CandRegs, NodeRegs) are
    // read with v_readlane from lanes that a divergent branch could have switched off, and a register the
    // compiler spills and reloads inside such a branch is restored for the active lanes only.
    const double absorbed_cost = (ab == 1) ? Model::boundcost(A.prm, x) : Model::obscost(A.prm, x);
    if (!__any(ab == 0)) return absorbed_cost; // wave-uniform: every lane absorbed
    typename Model::Node nd;
    Model::prep(A.prm, x, tv, nd);
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
        for (int i = 0; i < DU; i++) u[i] = cr.get_u(i, 0);
        double cf0[Model::NCF > 0 ? Model::NCF : 1];
        cf0[0] = 0.0;
#pragma unroll
        for (int i = 0; i < Model::NCF; i++) cf0[i] = cr.get_cf(i, 0);
        Model::drift(A.prm, nd, x, u, cf0, b);
        Model::sigma(A.prm, x, u, s);
        if constexpr (!Model::STAGE_UDEP) stage0 = Model::stage(A.prm, x, u);
        else if constexpr (stage_usep<Model>()) stage0 = Model::stage_x(A.prm, x); // + the candidate's stage_u below
#pragma unroll
        for (int m = 0; m < D; m++) {
            if (!((UM >> m) & 1u)) {
                double pm, pp;
                if ((Pre::MASK >> m) & 1u) { pm = pre.pm(m); pp = pre.pp(m); } // a constant of the fiber, formed once with upwind_rates
                else upwind_rates(A.t[2 * m], A.t[2 * m + 1], b[m], s[m], pm, pp);
                Q0 += pm;
                Q0 += pp;
                PV0 = fma(pm, V[2 * m], PV0);
                PV0 = fma(pp, V[2 * m + 1], PV0);
            }
        }
    }
    double best = 0.0;
    constexpr int NCFa = Model::NCF > 0 ? Model::NCF : 1;
    if (A.discount == 0.0) {
        // Undiscounted problems (beta = 0, exp(-beta dt) = 1): value_c = (h2*stage_c + PV_c)/Q_c + residue, so
        // the scan compares the candidates as fractions num_c/Q_c by cross-multiplication and divides ONCE for
        // the winner -- nine dependent f64 divisions per node are the longest chain of the scan otherwise.
        // Scan order and strict '<' are kept; a different winner is possible only between candidates whose
        // values agree to rounding.
        // best so far as the fraction bnum/bq; +inf loses to the first candidate that is not skipped
        double bnum = __builtin_inf(), bq = 1.0;
        bool anybad = false;
        constexpr unsigned UCm = Model::UCONST_MASK;
        constexpr bool ALLC = (UCm == UM) && !Model::STAGE_UDEP && Model::NCF == 0; // nothing per candidate needs x
        const double base0 = fma(h2l, stage0, PV0); // the candidate-independent part of every numerator (ALLC)
        // CGD candidates per trip: one candidate is a chain of ~10 dependent f64 operations and with two
        // wavefronts per SIMD nothing else covers their latency, so independent candidates are interleaved;
        // the selection below is straight-line (no short-circuit branches) and keeps the scan order.
        for (int c0 = 0; c0 < nc; c0 += CGD) {
            double Qq[CGD], numq[CGD];
#pragma unroll
            for (int q = 0; q < CGD; q++) {
                const int c = (CGD == 1 || c0 + q < nc) ? c0 + q : nc - 1;
                double Q, num;
                if constexpr (ALLC) {
                    // every control-dependent rate is a constant of the candidate: 2 FMAs per such dim and one add
                    double PVa = 0.0, PVb = 0.0;
#pragma unroll
                    for (int m = 0; m < D; m++) {
                        if ((UM >> m) & 1u) {
                            PVa = fma(cr.get_rpm(CandRegs<Model>::ucslot(m), c), V[2 * m], PVa);
                            PVb = fma(cr.get_rpp(CandRegs<Model>::ucslot(m), c), V[2 * m + 1], PVb);
                        }
                    }
                    Q = Q0 + cr.get_qab(c);
                    num = base0 + (PVa + PVb);
                } else {
                    double u[DU], cf[NCFa];
#pragma unroll
                    for (int i = 0; i < DU; i++) u[i] = cr.get_u(i, c);
                    cf[0] = 0.0;
#pragma unroll
                    for (int i = 0; i < Model::NCF; i++) cf[i] = cr.get_cf(i, c);
                    double b[D], sg[D];
                    Model::drift(A.prm, nd, x, u, cf, b);
                    Model::sigma(A.prm, x, u, sg);
                    double stage = stage0;
                    if constexpr (stage_usep<Model>()) stage = stage0 + cr.get_cf(0, c);
                    else if constexpr (Model::STAGE_UDEP) stage = Model::stage(A.prm, x, u);
                    double Qa = 0.0, Qb = 0.0, PVa = 0.0, PVb = 0.0; // two short chains instead of one long one
#pragma unroll
                    for (int m = 0; m < D; m++) {
                        if ((UM >> m) & 1u) {
                            double pm, pp;
                            if ((UCm >> m) & 1u) {
                                pm = cr.get_rpm(CandRegs<Model>::ucslot(m), c);
                                pp = cr.get_rpp(CandRegs<Model>::ucslot(m), c);
                            } else {
                                const double half = t2l[m] * (sg[m] * sg[m]) / 2.0;
                                const double tb = tl[m] * b[m];
                                pm = (b[m] < -1e-14) ? half - tb : half;
                                pp = (b[m] > 1e-14) ? half + tb : half;
                            }
                            if constexpr (UCm != UM) { // otherwise the sum of the candidate's rates is a table entry
                                Qa += pm;
                                Qb += pp;
                            }
                            PVa = fma(pm, V[2 * m], PVa);
                            PVb = fma(pp, V[2 * m + 1], PVb);
                        }
                    }
                    if constexpr (UCm == UM) Q = Q0 + cr.get_qab(c);
                    else Q = Q0 + (Qa + Qb);
                    num = fma(h2l, stage, PV0 + (PVa + PVb));
                }
                Qq[q] = Q;
                numq[q] = num;
            }
#pragma unroll
            for (int q = 0; q < CGD; q++) {
                if (CGD == 1 || c0 + q < nc) { // wave-uniform
                    const bool okc = !(Qq[q] < 1e-14); // nodeutil.c:365-367 returns 1; bellman.c:452 asserts.  Skip + flag.
                    anybad |= !okc;
                    double lhs = numq[q] * bq, rhs = bnum * Qq[q];
                    pin_vgpr(lhs); // evaluated unconditionally: the compiler must not wrap them in a divergent branch
                    pin_vgpr(rhs);
                    const bool better = lhs < rhs;
                    const bool take = okc & (forced ? (c0 + q == fu) : better);
                    bnum = take ? numq[q] : bnum;
                    bq = take ? Qq[q] : bq;
                    ui = take ? c0 + q : ui;
                }
            }
        }
        if (anybad & (ab == 0)) st |= C3SC_STATUS_STATIONARY;
        {
            // bq is a sum of rates in [1e-14, ~1e8] (or 1.0 when no candidate was valid): the hardware seed with two Newton steps
            // is correctly rounded over 4M random samples (tools/probes/probe_rcp.hip) and 5 instructions against the IEEE
            // sequence's 13
            const double inv = rcp_newton(bq);
            const double pself = fma(-bq, inv, 1.0);
            best = (ui >= 0) ? fma(pself, V[2 * D], bnum * inv) : 0.0;
        }
        best = (ab != 0) ? absorbed_cost : best;
        ui = (ab != 0) ? -1 : ui;
        return best;
    }
    // Candidates are evaluated CG at a time (CG = 1 where registers are tight): one candidate is a ~40-deep
    // chain of dependent f64 operations (rates -> Q -> 1/Q -> dt -> value), and with one or two wavefronts per
    // SIMD nothing else hides that latency, so independent candidates are interleaved.
    double bestg = __builtin_inf();
    bool anybad_g = false;
    // every candidate only adds rates to Q0, so dt_c <= h2/Q0: if beta*h2/Q0 is small on every lane, the discount
    // factor of every candidate takes the polynomial and the per-candidate wave vote is not needed
    const bool all_small = __all(discl * h2l < 0.0078125 * Q0);
    const bool all_tiny = __all(discl * h2l < 0.0009765625 * Q0); // 2^-10: four terms are exact to an ulp
    // ... and for the same reason Q_c >= Q0: when no lane's control-independent rates sum to less than 1e-14, no candidate can
    // trip the stationary test (nodeutil.c:365-367) and the scan runs without it
    const bool q0ok = __all(!(Q0 < 1e-14));
    // The scan body is instantiated per (stationary test on/off, form of the discount factor) and the choice is made ONCE per
    // node, outside the loop (all four conditions are wave-uniform): as run-time tests inside the body they cost every candidate a
    // compare, two selects and ~8 scalar instructions and branches -- a quarter of a 33-candidate scan (lqg2d: 30 VALU and 13
    // SALU instructions per candidate before, 26 and 5 after).
    auto scan = [&](auto check_tag, auto exp_tag) __attribute__((always_inline)) {
        constexpr bool CHECK = decltype(check_tag)::value;
        constexpr int EXPM = decltype(exp_tag)::value; // 0: four-term polynomial, 1: degree 7, 2: per-candidate vote (libm path possible), 3: any of them, chosen here
        for (int c0 = 0; c0 < nc; c0 += CG) {
            double val[CG];
            bool ok[CG];
#pragma unroll
            for (int q = 0; q < CG; q++) {
                const int c = (CG == 1 || c0 + q < nc) ? c0 + q : nc - 1;
                double u[DU], cf[NCFa];
#pragma unroll
                for (int i = 0; i < DU; i++) u[i] = cr.get_u(i, c);
                cf[0] = 0.0;
#pragma unroll
                for (int i = 0; i < Model::NCF; i++) cf[i] = cr.get_cf(i, c);
                double b[D], s[D];
                Model::drift(A.prm, nd, x, u, cf, b);
                Model::sigma(A.prm, x, u, s);
                double stage = stage0;
                if constexpr (stage_usep<Model>()) stage = stage0 + cr.get_cf(0, c);
                else if constexpr (Model::STAGE_UDEP) stage = Model::stage(A.prm, x, u);
                double Q = Q0, PV = PV0;
                constexpr unsigned UCg = Model::UCONST_MASK;
#pragma unroll
                for (int m = 0; m < D; m++) {
                    if ((UM >> m) & 1u) {
                        double pm, pp;
                        if ((UCg >> m) & 1u) { // the rates of this dim are constants of the candidate (table built with the candidates)
                            pm = cr.get_rpm(CandRegs<Model>::ucslot(m), c);
                            pp = cr.get_rpp(CandRegs<Model>::ucslot(m), c);
                        } else {
                            const double half = t2l[m] * (s[m] * s[m]) / 2.0;
                            // branch-free upwinding with the +-1e-14 dead zone of nodeutil.c:300-305
                            const double tb = tl[m] * b[m];
                            pm = (b[m] < -1e-14) ? half - tb : half;
                            pp = (b[m] > 1e-14) ? half + tb : half;
                        }
                        if constexpr (UCg != UM) { // otherwise the sum of the candidate's rates is a table entry
                            Q += pm;
                            Q += pp;
                        }
                        PV = fma(pm, V[2 * m], PV);
                        PV = fma(pp, V[2 * m + 1], PV);
                    }
                }
                if constexpr (UCg == UM) Q = Q0 + cr.get_qab(c);
                double Qs = Q;
                ok[q] = true;
                if constexpr (CHECK) {
                    ok[q] = !(Q < 1e-14); // nodeutil.c:365-367 returns 1; bellman.c:452 asserts.  Skip + flag.
                    Qs = ok[q] ? Q : 1.0;
                }
                const double inv = rcp_newton(Qs);         // Q in [1e-14, ~1e8]: no scaling / fix-up needed, result within an ulp
                const double dt = h2l * inv;               // nodeutil.c:369
                const double pself = fma(-Qs, inv, 1.0);   // 1 - sum_i p_i/Q: rounding residue, as in the reference
                const double ctg = fma(pself, V[2 * D], PV * inv);
                const double xe = -discl * dt;
                const double ebt = EXPM == 0 ? exp_tiny(xe) : (EXPM == 1 ? exp_small(xe) : (EXPM == 2 ? exp_discount(xe) : // bellman.c:94
                                   (all_tiny ? exp_tiny(xe) : (all_small ? exp_small(xe) : exp_discount(xe)))));
                val[q] = dt * stage + ebt * ctg;                                 // bellman.c:97
            }
#pragma unroll
            for (int q = 0; q < CG; q++) {
                if (c0 + q < nc) {
                    if constexpr (CHECK) anybad_g |= !ok[q];
                    if (__builtin_constant_p(forced) && !forced && !CHECK) { // the minimising kernels' fast path: compare, min, one select
                        ui = (val[q] < bestg) ? c0 + q : ui;
                        // v_min_f64 by hand: fmin() canonicalises the loop-carried operand first (a v_max_f64 x, x per candidate)
                        asm("v_min_f64 %0, %1, %2" : "=v"(bestg) : "v"(val[q]), "0"(bestg));
                    } else {
                        const bool take = ok[q] & (forced ? (c0 + q == fu) : (val[q] < bestg)); // +inf loses to the first candidate
                        bestg = take ? val[q] : bestg;
                        ui = take ? c0 + q : ui;
                    }
                }
            }
        }
    };
    typedef std::integral_constant<int, 0> E0;
    typedef std::integral_constant<int, 1> E1;
    typedef std::integral_constant<int, 2> E2;
    typedef std::integral_constant<int, 3> E3; // the form is chosen inside the body (one copy of the scan)
    // SPLIT = false: kernels at their register limit (the quad kernels) keep ONE scan body -- four copies of it perturb their
    // allocation by 2-3 % even where the discounted branch never runs (scar4d), and their candidate lists are short
    if constexpr (!SPLIT) scan(std::true_type{}, E3{});
    else if (q0ok) {
        if (all_tiny) scan(std::false_type{}, E0{});
        else if (all_small) scan(std::false_type{}, E1{});
        else scan(std::false_type{}, E2{});
    } else
        scan(std::true_type{}, E2{});
    if (anybad_g & (ab == 0)) st |= C3SC_STATUS_STATIONARY;
    best = (ui >= 0) ? bestg : 0.0;
    best = (ab != 0) ? absorbed_cost : best;
    ui = (ab != 0) ? -1 : ui;
    return best;
}

// Continuous control in a box [ulb, uub]^DU.  The reference hands bellman_control to C3's BFGS with a multistart
// schedule (box centre, quarter points, random restarts, bellman.c:545-1118) -- an optimiser that lives in C3 and is
// unseeded for du >= 2, so nothing about its trajectory is pinned.  Here: a dense tensor grid of
// `ugrid` points per control dimension (wave-uniform candidates generated arithmetically) followed by `upolish`
// rounds of per-lane coordinate golden-section search inside the cell around the best grid point.  Fixed trip
// counts and selects only; the objective is exactly node_backup's (rates, dt, bellmanrhs).  Models with per-candidate
// features (NCF > 0) must be able to form them from u on the device (Model::CF_FROM_U, Model::features): Cothrust6D does, with the
// device's sin / cos; Scar4D (tan u0) is not served in this mode.
template <class Model>
__device__ inline double node_backup_box(const KArgs &A, const double (&x)[Model::D],
                                         const double (&tv)[Model::NTAB > 0 ? Model::NTAB : 1],
                                         const double (&V)[2 * Model::D + 1], int ab, double (&uo)[Model::DU], unsigned &st,
                                         bool forced, const double *fu)
{
    constexpr int D = Model::D, DU = Model::DU;
    static_assert(DU <= C3SC_MAX_DU, "control dimension");
#pragma unroll
    for (int i = 0; i < DU; i++) uo[i] = 0.0;
    if (ab == 1) return Model::boundcost(A.prm, x);
    if (ab == -1) return Model::obscost(A.prm, x);
    typename Model::Node nd;
    Model::prep(A.prm, x, tv, nd);
    constexpr unsigned UM = Model::UDEP_MASK;
    constexpr int NCFb = Model::NCF > 0 ? Model::NCF : 1;
    double cf0[NCFb];
#pragma unroll
    for (int i = 0; i < NCFb; i++) cf0[i] = 0.0;
    double Q0 = 0.0, PV0 = 0.0;
    {
        double u0[DU], b[D], s[D];
#pragma unroll
        for (int i = 0; i < DU; i++) u0[i] = A.ulb[i];
        Model::drift(A.prm, nd, x, u0, cf0, b);
        Model::sigma(A.prm, x, u0, s);
#pragma unroll
        for (int m = 0; m < D; m++)
            if (!((UM >> m) & 1u)) {
                const double half = A.t[2 * m + 1] * (s[m] * s[m]) / 2.0;
                const double tb = A.t[2 * m] * b[m];
                const double pm = (b[m] < -1e-14) ? half - tb : half;
                const double pp = (b[m] > 1e-14) ? half + tb : half;
                Q0 += pm;
                Q0 += pp;
                PV0 = fma(pm, V[2 * m], PV0);
                PV0 = fma(pp, V[2 * m + 1], PV0);
            }
    }
    bool any_stationary = false;
    auto evalu = [&](const double (&u)[DU]) -> double {
        double b[D], s[D];
        if constexpr (Model::NCF > 0) { // features of a continuous control: formed from u here (models that can: CF_FROM_U)
            double cfu[NCFb];
            Model::features(u, cfu);
            Model::drift(A.prm, nd, x, u, cfu, b);
        } else
            Model::drift(A.prm, nd, x, u, cf0, b);
        Model::sigma(A.prm, x, u, s);
        const double stage = Model::stage(A.prm, x, u);
        double Q = Q0, PV = PV0;
#pragma unroll
        for (int m = 0; m < D; m++)
            if ((UM >> m) & 1u) {
                const double half = A.t[2 * m + 1] * (s[m] * s[m]) / 2.0;
                const double tb = A.t[2 * m] * b[m];
                const double pm = (b[m] < -1e-14) ? half - tb : half;
                const double pp = (b[m] > 1e-14) ? half + tb : half;
                Q += pm;
                Q += pp;
                PV = fma(pm, V[2 * m], PV);
                PV = fma(pp, V[2 * m + 1], PV);
            }
        const bool ok = !(Q < 1e-14);
        any_stationary |= !ok;
        const double Qs = ok ? Q : 1.0;
        const double inv = 1.0 / Qs;
        const double dt = A.h2 * inv;
        const double pself = fma(-Qs, inv, 1.0);
        const double ctg = fma(pself, V[2 * D], PV * inv);
        const double ebt = (A.discount == 0.0) ? 1.0 : exp_discount(-A.discount * dt);
        const double val = dt * stage + ebt * ctg;
        return ok ? val : 1.0e300;
    };
    double ub[DU], best;
    if (forced) { // policy evaluation: apply the given control
#pragma unroll
        for (int i = 0; i < DU; i++) ub[i] = fu[i];
        best = evalu(ub);
    } else {
        double dl[DU];
        const int G = A.ugrid;
        int total = 1;
#pragma unroll
        for (int i = 0; i < DU; i++) { dl[i] = (G > 1) ? (A.uub[i] - A.ulb[i]) / (double)(G - 1) : 0.0; total *= G; }
        best = 1.0e301;
#pragma unroll
        for (int i = 0; i < DU; i++) ub[i] = A.ulb[i];
        for (int c = 0; c < total; c++) { // wave-uniform candidates
            double u[DU];
            int rem = c;
#pragma unroll
            for (int i = 0; i < DU; i++) { const int gi = rem % G; rem /= G; u[i] = (gi == G - 1) ? A.uub[i] : fma((double)gi, dl[i], A.ulb[i]); }
            const double v = evalu(u);
            const bool take = v < best;
            best = take ? v : best;
#pragma unroll
            for (int i = 0; i < DU; i++) ub[i] = take ? u[i] : ub[i];
        }
        const double gr = 0.6180339887498949;
        for (int round = 0; round < A.upolish; round++) {
#pragma unroll
            for (int i = 0; i < DU; i++) {
                double lo = fmax(A.ulb[i], ub[i] - dl[i]), hi = fmin(A.uub[i], ub[i] + dl[i]);
                double u[DU];
#pragma unroll
                for (int q = 0; q < DU; q++) u[q] = ub[q];
                double x1 = hi - gr * (hi - lo), x2 = lo + gr * (hi - lo);
                u[i] = x1;
                double f1 = evalu(u);
                u[i] = x2;
                double f2 = evalu(u);
                for (int it = 0; it < 40; it++) {
                    const bool left = f1 < f2; // keep [lo, x2]
                    hi = left ? x2 : hi;
                    lo = left ? lo : x1;
                    const double xn = left ? hi - gr * (hi - lo) : lo + gr * (hi - lo);
                    u[i] = xn;
                    const double fn = evalu(u);
                    const double ox1 = x1, of1 = f1;
                    x1 = left ? xn : x2;
                    f1 = left ? fn : f2;
                    x2 = left ? ox1 : xn;
                    f2 = left ? of1 : fn;
                }
                const double xm = (f1 < f2) ? x1 : x2, fm = fmin(f1, f2);
                const bool take = fm < best;
                best = take ? fm : best;
                ub[i] = take ? xm : ub[i];
            }
        }
    }
    if (any_stationary) st |= C3SC_STATUS_STATIONARY;
#pragma unroll
    for (int i = 0; i < DU; i++) uo[i] = ub[i];
    return best;
}

} // namespace c3sc
