// kernel_fiber_pair.hpp -- "one fiber per lane, two wavefronts per 64 fibers" Bellman kernel (gfx950).
//
// Fold-once algebra (fold everything that is constant along a fiber into
// L, R and the 2(d-1) neighbour vectors w_m^{-+} / z_m^{-+}; per node only c = G_k[j] R, a = L G_k[j]
// and 2d-1 short dots remain), but the folded vectors do not fit the 256 VALU-addressable VGPRs of a
// lane (2(d-1) r doubles = 240 VGPRs at d = 7, r = 10).  So a WORKGROUP OF TWO WAVEFRONTS owns 64
// fibers and the rank index is split in halves: wave h holds components [h r/2, (h+1) r/2) of every
// neighbour vector (60 doubles), computes the matching half of c and a with the wave-uniform matrix
// element as an SGPR operand (both waves see the same j, their element offsets differ by a wave-uniform
// constant), and produces PARTIAL dots; the two waves exchange the 2d-1 partial sums per node through
// LDS and take turns running the control minimisation (wave 0: even nodes, wave 1: odd nodes, delayed
// by one node pair so the dim-k neighbours are known).  One s_barrier pair per two nodes.
//
// Folding (once per 64-fiber tile): each wave folds the neighbour vectors of half of the dims (full
// length, 6 x r doubles in registers) through the fixed cores staged in LDS -- exactly as the
// fiber-per-lane kernel does -- and the halves are swapped through LDS afterwards.
//
// Budget: 128-thread workgroups, <= 256 VGPRs (2 waves per SIMD with 4 workgroups per CU), LDS = one
// staged core (33 KB at N = 41, r = 10), reused as the exchange buffer during the node loop.
#pragma once
#include <utility>

#include "fold_lds.hpp"

#ifndef FPP_CGD
#define FPP_CGD 3
#endif
#ifndef FPP_PIPE2
#define FPP_PIPE2 1
#endif
#ifndef FPP_CG
#define FPP_CG 1 // candidates in flight in the discounted scan (one division + exp chain each); 1 where registers are tight
#endif
#ifndef FPP_NV
#define FPP_NV 4 // vectors per pass over a staged matrix in the folding phase (2: 0.361 ms, 3: 0.343, 4: 0.340 on car7d)
#endif

#ifndef FPP_DIRECT_MAXD
#define FPP_DIRECT_MAXD 3
#endif

namespace c3sc {

constexpr int FPP_THREADS = 128;

// Fold straight from the cores in global memory (fold_lds.hpp: apply_glb) instead of through a staged copy: a 3-D problem
// folds through ONE matrix level at most, and the staged core (30 KB for dubins3d, per 64 fibers) is what holds its
// workgroups at two wavefronts per SIMD -- without it the exchange rows are the LDS footprint (21 KB: three to four).
template <class Model, int RP>
__host__ __device__ constexpr bool fpp_direct() { return Model::D <= FPP_DIRECT_MAXD && RP <= 8; }

// apply the staged matrix G to W[FIRST .. FIRST+COUNT) AND to one extra vector E in the same passes: every pass
// re-reads the 100-element matrix of the lane's node from LDS, and LDS (gathered rows, ~2x bank conflicts) is the
// busiest unit of the folding phase, so the running prefix/suffix rides along instead of taking a pass of its own
template <int RP, int NW, int FIRST, int COUNT, bool ROWVEC>
__device__ inline void apply_core_and(const double *G, double (&W)[NW][RP], double (&E)[RP])
{
    constexpr int N1 = COUNT >= FPP_NV - 1 ? FPP_NV - 1 : COUNT;
    double tmp[N1 + 1][RP];
#pragma unroll
    for (int a = 0; a < RP; a++) tmp[0][a] = E[a];
#pragma unroll
    for (int s = 0; s < N1; s++)
#pragma unroll
        for (int a = 0; a < RP; a++) tmp[s + 1][a] = W[FIRST + s][a];
    if constexpr (ROWVEC) vecmat_lds<RP, N1 + 1>(G, tmp);
    else matvec_lds<RP, N1 + 1>(G, tmp);
#pragma unroll
    for (int a = 0; a < RP; a++) E[a] = tmp[0][a];
#pragma unroll
    for (int s = 0; s < N1; s++)
#pragma unroll
        for (int a = 0; a < RP; a++) W[FIRST + s][a] = tmp[s + 1][a];
    apply_core<RP, NW, FIRST + N1, COUNT - N1, ROWVEC, FPP_NV>(G, W);
}

__device__ inline void pair_barrier()
{ // workgroup barrier + LDS visibility between the two wavefronts.  Only LDS traffic is exchanged, so only
  // lgkmcnt is drained: a workgroup-scope release fence would also wait (vmcnt(0)) for the scattered
  // output stores of the previous node, which costs microseconds per barrier.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One fixed core into LDS by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS at wave base + 16 lane).
// The source is the core's LDS image in HBM (padded node stride, k_core_image), so the copy is lane-linear.  No VGPRs and
// every piece of the core in flight at once -- the register-staged copy it replaces took three rounds of L2 latency per core
// with eight 16-byte loads per thread in flight.
// `img` must come from KArgs::img_base, not from the kernel's `ro` argument: the intrinsic counts as a memory write, and once a
// pointer based on the __restrict__ `ro` has been handed to it every later load from `ro` is "possibly clobbered" -- the
// wave-uniform core rows of the node loop would stop being scalar loads.
template <int H>
__device__ inline void stage_core_image(double *sK, const double *img, int n_doubles)
{
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int pieces = (n_doubles + 1) >> 1;
    const int lane = threadIdx.x & 63;
    // A wave-instruction writes 64 consecutive pieces.  The last round is moved back so that it ends with the image (it
    // re-copies a few pieces: same data to the same place) instead of switching lanes off -- the kernel keeps EXEC untouched
    // (tests/test_kernel_uniform_control_flow.py).  An image of fewer than 64 pieces is followed by repeats of its last piece;
    // the staging area is never smaller than the 11 KB exchange block, so those stay inside it.
    for (int base = H * 64; base < pieces; base += FPP_THREADS) {
        const int b = max(min(base, pieces - 64), 0);
        const int p = min(b + lane, pieces - 1);
        __builtin_amdgcn_global_load_lds((glb_void *)(img + 2 * p), (lds_void *)((lds_char *)sK + 16 * b), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// which wave folds the neighbour pair of dim m (m != K): alternate by distance from K so the O(distance)
// propagation work is balanced
template <int K>
__host__ __device__ constexpr int pair_owner(int m) { return m < K ? ((K - 1 - m) & 1) : ((m - K) & 1); }

template <int D, int K, int H>
__host__ __device__ constexpr int own_left_before(int m)
{ // number of left dims < m owned by H
    int c = 0;
    for (int q = 0; q < m && q < K; q++) c += (pair_owner<K>(q) == H);
    return c;
}
template <int D, int K, int H>
__host__ __device__ constexpr int own_right_after(int m)
{ // number of right dims > m owned by H
    int c = 0;
    for (int q = D - 1; q > m && q > K; q--) c += (pair_owner<K>(q) == H);
    return c;
}

// dims-neighbour vector index (skips K): (m, s) -> 0 .. 2(D-1)-1
template <int K>
__host__ __device__ constexpr int gvec(int m, int s) { return (m < K ? 2 * m : 2 * (m - 1)) + s; }

// What a fiber parks in LDS for the moment its nodes are finalised (rows of 64 lanes behind the exchange rows).  Without
// dependency information (Model::HAS_DEPS absent): every fixed coordinate and every table value.  With it: the upwind rates of
// the control-independent dimensions whose drift does not read the varying dimension are constants of the fiber -- formed once
// (upwind_rates, the node loop's own arithmetic) and parked INSTEAD of the coordinates / table values only they needed; the node
// then reads two doubles per such dimension where it formed a drift, two compares, four selects and three flops (car7d: 3.7 of
// the 5 control-independent dimensions on average over K).
template <class Model, int K>
struct PairPark {
    static constexpr int D = Model::D, NTAB = Model::NTAB;
    static constexpr unsigned UM = Model::UDEP_MASK, UC = Model::UCONST_MASK, ALL = (1u << D) - 1u;
    __host__ __device__ static constexpr bool has_deps()
    {
        if constexpr (requires { Model::HAS_DEPS; }) return Model::HAS_DEPS;
        else return false;
    }
    __host__ __device__ static constexpr unsigned dep(int m)
    {
        if constexpr (has_deps()) return Model::dep_mask(m);
        else return ALL;
    }
    __host__ __device__ static constexpr unsigned cost_dep()
    {
        if constexpr (has_deps()) return Model::COST_DEP;
        else return ALL;
    }
    // control-independent dimensions (other than K... K itself may be one too) whose rates are constants of the fiber
    __host__ __device__ static constexpr unsigned constd()
    {
        unsigned c = 0;
        if (has_deps())
            for (int m = 0; m < D; m++)
                if (!((UM >> m) & 1u) && !((dep(m) >> K) & 1u)) c |= 1u << m;
        return c;
    }
    // state dimensions whose coordinate (or tables) a node still needs: the costs', the non-constant control-independent drifts',
    // the state-dependent controlled drifts'
    __host__ __device__ static constexpr unsigned need()
    {
        unsigned n = cost_dep();
        for (int m = 0; m < D; m++) {
            const bool ctl = (UM >> m) & 1u, cst = (constd() >> m) & 1u;
            if ((!ctl && !cst) || (ctl && !((UC >> m) & 1u))) n |= dep(m);
        }
        return n & ALL;
    }
    __host__ __device__ static constexpr bool need_x(int m) { return m != K && ((need() >> m) & 1u); }
    __host__ __device__ static constexpr bool need_t(int t) { return Model::tab_dim(t) != K && ((need() >> Model::tab_dim(t)) & 1u); }
    __host__ __device__ static constexpr int row_x(int m)
    {
        int r = 0;
        for (int q = 0; q < m; q++) r += need_x(q);
        return r;
    }
    __host__ __device__ static constexpr int nx() { return row_x(D); }
    __host__ __device__ static constexpr int row_t(int t)
    {
        int r = nx();
        for (int q = 0; q < t; q++) r += need_t(q);
        return r;
    }
    __host__ __device__ static constexpr int nt() { return row_t(NTAB) - nx(); }
    __host__ __device__ static constexpr int row_pm(int m)
    {
        int r = nx() + nt();
        for (int q = 0; q < m; q++) r += 2 * ((constd() >> q) & 1u);
        return r;
    }
    __host__ __device__ static constexpr int rows() { return row_pm(D); }
};

// the parked rates as node_backup's Pre
template <class Model, int K>
struct PairPre {
    static constexpr unsigned MASK = PairPark<Model, K>::constd();
    const double *PX;
    int lane;
    __device__ inline double pm(int m) const { return PX[PairPark<Model, K>::row_pm(m) * 64 + lane]; }
    __device__ inline double pp(int m) const { return PX[(PairPark<Model, K>::row_pm(m) + 1) * 64 + lane]; }
};

#define FPP_STAMP(slot)                                                   \
    if (C3SC_STAMPS_ON && (A.dbg & 128)) {                                \
        const unsigned long long now__ = clock64();                       \
        seg[slot] += now__ - tlast;                                       \
        tlast = now__;                                                    \
    }

template <class Model, int RP, int K, int H, bool FORCED>
__device__ __attribute__((always_inline)) inline void fiber_pair_body(const KArgs &A, const double *__restrict__ ro,
                                                                     const int32_t *__restrict__ idx, double *__restrict__ outv,
                                                                     int32_t *__restrict__ uidx, int32_t *__restrict__ absorbed,
                                                                     double *sK, unsigned &st)
{
    constexpr int D = Model::D;
    constexpr int S = 2 * D + 1;
    constexpr int RH = RP / 2;
    constexpr int NV = 2 * (D - 1);                      // neighbour vectors in total
    constexpr int NOL = 2 * own_left_before<D, K, H>(K); // own left vectors (slots [0, NOL))
    constexpr int NOR = 2 * own_right_after<D, K, H>(K); // own right vectors (slots [NOL, NOL+NOR))
    constexpr int NOWN = (NOL + NOR) > 0 ? (NOL + NOR) : 1;
    constexpr int NP = NV + 1; // partial sums per node: NV neighbour values + the node value
    constexpr bool DIRECT = fpp_direct<Model, RP>();
    static_assert(RP % 2 == 0, "rank-split kernel needs an even padded rank");
    const int lane = threadIdx.x & 63;
    const int N = A.N;
    const long ntiles = (A.F + 63) / 64;
    unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
    // candidate and node tables: computed lane-distributed once, then kept in LDS behind the exchange rows and read
    // with wave-uniform addresses -- 16 VGPRs less in the node loop than keeping them in lanes, and no dependence on
    // what a spill does to inactive lanes
    CandLds<Model> cr;
    NodeLds<Model, K> nr;
    {
        CandRegs<Model> cr0;
        cr0.load(A, ro);
        NodeRegs<Model, K> nr0;
        nr0.load(A, ro);
        double *tb = sK + A.tbl_off;
        cr.fill(tb, cr0, A.ncand);
        nr.fill(tb + CandLds<Model>::doubles(A.ncand), nr0, N);
        pair_barrier();
    }

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        FPP_STAMP(7)
        const long f_raw = tile * 64 + lane;
        const bool live = f_raw < A.F;
        const long f = live ? f_raw : A.F - 1;

        int fi[D], nbm[D], nbp[D];
        bool fiber_abs = false;
        double x[D];
#pragma unroll
        for (int m = 0; m < D; m++) {
            fi[m] = (m == K) ? 0 : idx[f * D + m];
            const bool face = fixed_neighbors(fi[m], A.ngrid[m], A.bctype[m], nbm[m], nbp[m]);
            if (m != K) fiber_abs = fiber_abs || face;
            x[m] = ro[A.xg_off[m] + fi[m]];
        }

        const unsigned obs_fixed = obstacle_mask_fixed<D>(A, ro, x, K);
        double tv[Model::NTAB > 0 ? Model::NTAB : 1]; // model tables: constant along the fiber unless indexed by dim K
        table_values<Model>(A, ro, fi, tv);

        double L[RP], R[RP], W[NOWN][RP];
#pragma unroll
        for (int a = 0; a < RP; a++) { L[a] = (a == 0) ? 1.0 : 0.0; R[a] = (a == 0) ? 1.0 : 0.0; }

        // ------------------------------------------------------------ fold the prefix side
        FPP_STAMP(0) // tile setup
        if (!(C3SC_STAMPS_ON && (A.dbg & 1))) {
        if constexpr (K > 0) {
            {
                constexpr int str = DIRECT ? RP : fpl_lds_stride(RP);
                const double *src = sK;
                FPP_STAMP(1)
                if constexpr (DIRECT) src = ro + A.core_off[0];
                else {
                    pair_barrier();
                    stage_core_image<H>(sK, A.img_base + A.pair_img_off[0], A.ngrid[0] * str);
                    pair_barrier();
                }
                FPP_STAMP(7)
#pragma unroll
                for (int b = 0; b < RP; b++) {
                    L[b] = src[fi[0] * str + b];
                    if constexpr (pair_owner<K>(0) == H) {
                        W[0][b] = src[nbm[0] * str + b];
                        W[1][b] = src[nbp[0] * str + b];
                    }
                }
            }
            auto left_step = [&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                constexpr int str = DIRECT ? RP * RP : fpl_lds_stride(RP * RP);
                constexpr int before = 2 * own_left_before<D, K, H>(m); // own vectors created so far
                const double *src = sK;
                FPP_STAMP(1)
                if constexpr (DIRECT) src = ro + A.core_off[m];
                else {
                    pair_barrier();
                    stage_core_image<H>(sK, A.img_base + A.pair_img_off[m], A.ngrid[m] * str);
                    pair_barrier();
                }
                FPP_STAMP(7)
                const double *G = src + fi[m] * str;
                if constexpr (pair_owner<K>(m) == H) { // the new pair first: it needs the prefix BEFORE this core
                    double t0[1][RP], t1[1][RP];
#pragma unroll
                    for (int a = 0; a < RP; a++) { t0[0][a] = L[a]; t1[0][a] = L[a]; }
                    if constexpr (DIRECT) {
                        apply_glb<RP, 1, true>(src + nbm[m] * str, t0);
                        apply_glb<RP, 1, true>(src + nbp[m] * str, t1);
                    } else {
                        vecmat_lds<RP, 1>(src + nbm[m] * str, t0);
                        vecmat_lds<RP, 1>(src + nbp[m] * str, t1);
                    }
#pragma unroll
                    for (int a = 0; a < RP; a++) { W[before][a] = t0[0][a]; W[before + 1][a] = t1[0][a]; }
                }
                if constexpr (DIRECT) {
                    double tl[1][RP];
#pragma unroll
                    for (int a = 0; a < RP; a++) tl[0][a] = L[a];
                    apply_glb<RP, 1, true>(G, tl);
#pragma unroll
                    for (int a = 0; a < RP; a++) L[a] = tl[0][a];
                    apply_range_glb<RP, NOWN, 0, before, true>(G, W);
                } else
                    apply_core_and<RP, NOWN, 0, before, true>(G, W, L);
            };
            [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (left_step(std::integral_constant<int, Ms + 1>{}), ...); }
            (std::make_integer_sequence<int, (K > 1 ? K - 1 : 0)>{});
        }

        // ------------------------------------------------------------ fold the suffix side
        if constexpr (K < D - 1) {
            {
                constexpr int str = DIRECT ? RP : fpl_lds_stride(RP);
                const double *src = sK;
                FPP_STAMP(1)
                if constexpr (DIRECT) src = ro + A.core_off[D - 1];
                else {
                    pair_barrier();
                    stage_core_image<H>(sK, A.img_base + A.pair_img_off[D - 1], A.ngrid[D - 1] * str);
                    pair_barrier();
                }
                FPP_STAMP(7)
#pragma unroll
                for (int a = 0; a < RP; a++) {
                    R[a] = src[fi[D - 1] * str + a];
                    if constexpr (pair_owner<K>(D - 1) == H) {
                        W[NOL][a] = src[nbm[D - 1] * str + a];
                        W[NOL + 1][a] = src[nbp[D - 1] * str + a];
                    }
                }
            }
            auto right_step = [&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value; // D-2 down to K+1
                constexpr int str = DIRECT ? RP * RP : fpl_lds_stride(RP * RP);
                constexpr int after = 2 * own_right_after<D, K, H>(m);
                const double *src = sK;
                FPP_STAMP(1)
                if constexpr (DIRECT) src = ro + A.core_off[m];
                else {
                    pair_barrier();
                    stage_core_image<H>(sK, A.img_base + A.pair_img_off[m], A.ngrid[m] * str);
                    pair_barrier();
                }
                FPP_STAMP(7)
                const double *G = src + fi[m] * str;
                if constexpr (pair_owner<K>(m) == H) {
                    double t0[1][RP], t1[1][RP];
#pragma unroll
                    for (int a = 0; a < RP; a++) { t0[0][a] = R[a]; t1[0][a] = R[a]; }
                    if constexpr (DIRECT) {
                        apply_glb<RP, 1, false>(src + nbm[m] * str, t0);
                        apply_glb<RP, 1, false>(src + nbp[m] * str, t1);
                    } else {
                        matvec_lds<RP, 1>(src + nbm[m] * str, t0);
                        matvec_lds<RP, 1>(src + nbp[m] * str, t1);
                    }
#pragma unroll
                    for (int a = 0; a < RP; a++) { W[NOL + after][a] = t0[0][a]; W[NOL + after + 1][a] = t1[0][a]; }
                }
                if constexpr (DIRECT) {
                    double tr[1][RP];
#pragma unroll
                    for (int a = 0; a < RP; a++) tr[0][a] = R[a];
                    apply_glb<RP, 1, false>(G, tr);
#pragma unroll
                    for (int a = 0; a < RP; a++) R[a] = tr[0][a];
                    apply_range_glb<RP, NOWN, NOL, after, false>(G, W);
                } else
                    apply_core_and<RP, NOWN, NOL, after, false>(G, W, R);
            };
            [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (right_step(std::integral_constant<int, D - 2 - Ms>{}), ...); }
            (std::make_integer_sequence<int, (D - 2 - K > 0 ? D - 2 - K : 0)>{});
        }

        } // dbg & 1
        FPP_STAMP(1) // folding
        // ------------------------------------------------------------ swap halves: Wh[g][i] = component H*RH + i of vector g
        double Wh[NV][RH];
        {
            double *X = sK; // [NV][RH][64]
            pair_barrier();
            auto put_get_own = [&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                if constexpr (m != K && pair_owner<K>(m) == H) {
                    constexpr int slot = (m < K) ? 2 * own_left_before<D, K, H>(m) : NOL + 2 * own_right_after<D, K, H>(m);
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const int g = gvec<K>(m, s);
#pragma unroll
                        for (int i = 0; i < RH; i++) {
                            X[(g * RH + i) * 64 + lane] = W[slot + s][(1 - H) * RH + i];
                            Wh[g][i] = W[slot + s][H * RH + i];
                        }
                    }
                }
            };
            [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (put_get_own(std::integral_constant<int, Ms>{}), ...); }
            (std::make_integer_sequence<int, D>{});
            pair_barrier();
            auto get_other = [&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                if constexpr (m != K && pair_owner<K>(m) != H) {
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const int g = gvec<K>(m, s);
#pragma unroll
                        for (int i = 0; i < RH; i++) Wh[g][i] = X[(g * RH + i) * 64 + lane];
                    }
                }
            };
            [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (get_other(std::integral_constant<int, Ms>{}), ...); }
            (std::make_integer_sequence<int, D>{});
            pair_barrier();
        }
        // L and R move to LDS (rows of 64 lanes): both waves hold the same values, wave 0 stores L, wave 1 R.
        // The node loop re-reads them per use; that frees 2*RP*2 VGPRs for the 256-register budget.
        double *sL = sK, *sR = sK + RP * 64;
        {
#pragma unroll
            for (int a = 0; a < RP; a++) {
                if constexpr (H == 0) sL[a * 64 + lane] = L[a];
                else sR[a * 64 + lane] = R[a];
            }
            // the fiber's coordinates and table values are only needed when a node is finalised: parked in LDS
            // (wave 0 writes; both waves hold the same numbers) they do not occupy VGPRs during the partial sums
            if constexpr (H == 0) {
                typedef PairPark<Model, K> PK;
                static_assert(PK::rows() <= NP, "parking rows exceed the reserved block");
                double *PXw = sK + (2 * RP + 2 * (NP + 1) + NP) * 64;
#pragma unroll
                for (int m = 0; m < D; m++)
                    if (PK::need_x(m)) PXw[PK::row_x(m) * 64 + lane] = x[m];
#pragma unroll
                for (int t = 0; t < Model::NTAB; t++)
                    if (PK::need_t(t)) PXw[PK::row_t(t) * 64 + lane] = tv[t];
                if constexpr (PK::constd() != 0) {
                    // x[K] and the K-indexed tables are node 0's here (fi[K] = 0): the rates kept do not read them
                    typename Model::Node nd0;
                    Model::prep(A.prm, x, tv, nd0);
                    double u0[Model::DU], cf0[Model::NCF > 0 ? Model::NCF : 1], b0[D], s0[D];
#pragma unroll
                    for (int i = 0; i < Model::DU; i++) u0[i] = cr.get_u(i, 0);
                    cf0[0] = 0.0;
#pragma unroll
                    for (int i = 0; i < Model::NCF; i++) cf0[i] = cr.get_cf(i, 0);
                    Model::drift(A.prm, nd0, x, u0, cf0, b0);
                    Model::sigma(A.prm, x, u0, s0);
#pragma unroll
                    for (int m = 0; m < D; m++)
                        if ((PK::constd() >> m) & 1u) {
                            double pm, pp;
                            upwind_rates(A.t[2 * m], A.t[2 * m + 1], b0[m], s0[m], pm, pp);
                            PXw[PK::row_pm(m) * 64 + lane] = pm;
                            PXw[(PK::row_pm(m) + 1) * 64 + lane] = pp;
                        }
                }
            }
            pair_barrier();
        }
        FPP_STAMP(2) // half swap

        // ------------------------------------------------------------ node loop
        const int bck = A.bctype[K];
        const double *Gk = ro + A.core_off[K];
        // partial sums of node j owned by this wave, stored straight to LDS rows dst[g*64 + lane]:
        // g < NV neighbour values, g = NV the node value; the node-value partial is also returned
        auto partials = [&](int j, auto &&sink) __attribute__((always_inline)) -> double {
            double pv;
            if constexpr (K == 0) { // G_0[j] is a 1 x r row: a = row, no left vectors
                double ah[RH];
#pragma unroll
                for (int i = 0; i < RH; i++) ah[i] = Gk[(size_t)j * RP + H * RH + i];
                double v = 0.0;
#pragma unroll
                for (int i = 0; i < RH; i++) v = fma(ah[i], sR[(H * RH + i) * 64 + lane], v);
                pv = v;
#pragma unroll
                for (int g = 0; g < NV; g++) sink(g, dot_reg<RH>(ah, Wh[g]));
            } else if constexpr (K == D - 1) { // G_{d-1}[j] is an r x 1 column: c = column, no right vectors
                double ch[RH];
#pragma unroll
                for (int i = 0; i < RH; i++) ch[i] = Gk[(size_t)j * RP + H * RH + i];
                double v = 0.0;
#pragma unroll
                for (int i = 0; i < RH; i++) v = fma(sL[(H * RH + i) * 64 + lane], ch[i], v);
                pv = v;
#pragma unroll
                for (int g = 0; g < NV; g++) sink(g, dot_reg<RH>(Wh[g], ch));
            } else {
                // This wave's rows / columns of G_K[j] on the scalar path: c_h[i] = sum_b G[hi, b] R[b] (stride RP),
                // a_h[i] = sum_a L[a] G[a, hi] (contiguous).  A second, row-major copy of the core made the c-part
                // contiguous too and was 2.5 % faster in isolation, but it doubles what a workgroup pulls through the
                // 16 KB scalar cache (28 % of the requests missed or waited on a miss): without it 0.299 -> 0.282 ms.
                // (Walking the rows one at a time with the next row's s_load in flight was tried as well: SMEM
                // returns out of order, every wait is lgkmcnt(0), ten short waits lose to the compiler's bulk issue.)
                const double *GC0 = Gk + (size_t)j * RP * RP;
                const double *GC = GC0 + (size_t)H * RH * RP;
                double ch[RH], ah[RH];
#pragma unroll
                for (int i = 0; i < RH; i++) { ch[i] = 0.0; ah[i] = 0.0; }
#pragma unroll
                for (int b = 0; b < RP; b++) {
                    const double rb = sR[b * 64 + lane];
#pragma unroll
                    for (int i = 0; i < RH; i++) ch[i] = fma(GC0[H * RH + i + b * RP], rb, ch[i]);
                }
#pragma unroll
                for (int a = 0; a < RP; a++) {
                    const double la = sL[a * 64 + lane];
#pragma unroll
                    for (int i = 0; i < RH; i++) ah[i] = fma(la, GC[a + i * RP], ah[i]);
                }
                double v = 0.0;
#pragma unroll
                for (int i = 0; i < RH; i++) v = fma(sL[(H * RH + i) * 64 + lane], ch[i], v);
                pv = v;
#pragma unroll
                for (int g = 0; g < 2 * K; g++) sink(g, dot_reg<RH>(Wh[g], ch));
#pragma unroll
                for (int g = 2 * K; g < NV; g++) sink(g, dot_reg<RH>(ah, Wh[g]));
            }
            sink(NV, pv);
            return pv;
        };
        auto to_lds = [&](double *dst) __attribute__((always_inline)) {
            return [dst, lane](int g, double v) __attribute__((always_inline)) { dst[g * 64 + lane] = v; };
        };
        // The same for a middle core in two halves, so that two nodes can be software-pipelined: the scalar loads and
        // FMAs of the second node's (c, a) are in the instruction stream BEFORE the register-only dots of the first
        // node, which then cover the scalar-memory latency.
        auto ca_part = [&](int j, double (&ch)[RH], double (&ah)[RH], auto &&filler) __attribute__((always_inline)) {
            // Scalar loads return out of order, so every wait on one is lgkmcnt(0): nothing can stay in flight across a
            // wait.  Left alone the compiler loads just in time and waits ~8 times per node; here the 75 values a wave
            // needs are fetched in four batches of <= 25 doubles (50 SGPRs), each issued back to back and waited for once.
            const double *GC0 = Gk + (size_t)j * RP * RP;
            const double *GC = GC0 + (size_t)H * RH * RP;
#pragma unroll
            for (int i = 0; i < RH; i++) { ch[i] = 0.0; ah[i] = 0.0; }
            constexpr int BH = (RP + 1) / 2; // columns per batch
            double vec[RP]; // R, then L: read from LDS ahead of the scalar batch so that one wait covers both
#pragma unroll
            for (int b = 0; b < RP; b++) vec[b] = sR[b * 64 + lane];
#pragma unroll
            for (int b0 = 0; b0 < RP; b0 += BH) {
                double g[BH][RH];
#pragma unroll
                for (int b = 0; b < BH; b++)
#pragma unroll
                    for (int i = 0; i < RH; i++) g[b][i] = (b0 + b < RP) ? GC0[H * RH + i + (b0 + b) * RP] : 0.0;
                __builtin_amdgcn_sched_barrier(0); // loads stay above, the filler (register-only work) below
                filler(b0 / BH);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < BH; b++)
#pragma unroll
                    for (int i = 0; i < RH; i++) asm volatile("" : "+s"(g[b][i]));
#pragma unroll
                for (int b = 0; b < BH; b++) {
                    if (b0 + b < RP) {
                        const double rb = vec[b0 + b];
#pragma unroll
                        for (int i = 0; i < RH; i++) ch[i] = fma(g[b][i], rb, ch[i]);
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < RP; a++) vec[a] = sL[a * 64 + lane];
#pragma unroll
            for (int a0 = 0; a0 < RP; a0 += BH) {
                double g[RH][BH];
#pragma unroll
                for (int i = 0; i < RH; i++)
#pragma unroll
                    for (int a = 0; a < BH; a++) g[i][a] = (a0 + a < RP) ? GC[a0 + a + i * RP] : 0.0;
                __builtin_amdgcn_sched_barrier(0);
                filler(2 + a0 / BH);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < RH; i++)
#pragma unroll
                    for (int a = 0; a < BH; a++) asm volatile("" : "+s"(g[i][a]));
#pragma unroll
                for (int a = 0; a < BH; a++) {
                    if (a0 + a < RP) {
                        const double la = vec[a0 + a];
#pragma unroll
                        for (int i = 0; i < RH; i++) ah[i] = fma(la, g[i][a], ah[i]);
                    }
                }
            }
        };
        // quarter q (0..3) of the 2d-1 dots of a node: register-only work that fills the scalar-load waits of the next node
        auto dots_quarter = [&](int q, const double (&ch)[RH], const double (&ah)[RH], auto &&sink) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < NV; g++)
                if (g * 4 / NV == q) sink(g, (g < 2 * K) ? dot_reg<RH>(Wh[g], ch) : dot_reg<RH>(ah, Wh[g]));
        };
        auto dots_part = [&](const double (&ch)[RH], const double (&ah)[RH], auto &&sink) __attribute__((always_inline)) -> double {
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < RH; i++) v = fma(sL[(H * RH + i) * 64 + lane], ch[i], v);
#pragma unroll
            for (int g = 0; g < 2 * K; g++) sink(g, dot_reg<RH>(Wh[g], ch));
#pragma unroll
            for (int g = 2 * K; g < NV; g++) sink(g, dot_reg<RH>(ah, Wh[g]));
            sink(NV, v);
            return v;
        };
        // LDS exchange rows after L/R
        double *B0 = sK + 2 * RP * 64;    // wave 0 -> wave 1 : P_0(j1)[NP], then v_0(j0)
        double *B1 = B0 + (NP + 1) * 64;  // wave 1 -> wave 0 : P_1(j0)[NP], then v_1(j1)
        double *B2 = B1 + (NP + 1) * 64;  // wave 1 own       : P_1(j1)[NP]
        double *PX = B2 + NP * 64;        // parked per-fiber constants: x[m] (m != K) rows 0..D-1, model tables after

        // finalise one node from its assembled stencil (NV neighbour values in gvec order, node value last)
        auto finalize = [&](int jn, const double (&Vt)[NP], double vlo, double vhi) __attribute__((always_inline)) {
            double V[S];
#pragma unroll
            for (int m = 0; m < D; m++) {
                if (m == K) { V[2 * m] = vlo; V[2 * m + 1] = vhi; }
                else { V[2 * m] = Vt[gvec<K>(m, 0)]; V[2 * m + 1] = Vt[gvec<K>(m, 1)]; }
            }
            V[2 * D] = Vt[NV];
            double x[D], tv[Model::NTAB > 0 ? Model::NTAB : 1];
            tv[0] = 0.0;
#pragma unroll
            for (int m = 0; m < D; m++) // a coordinate nothing per node reads any more is not parked (PairPark)
                x[m] = (m == K) ? nr.x_at(jn) : (PairPark<Model, K>::need_x(m) ? PX[PairPark<Model, K>::row_x(m) * 64 + lane] : 0.0);
#pragma unroll
            for (int t = 0; t < Model::NTAB; t++)
                tv[t] = (Model::tab_dim(t) == K) ? nr.tab_at(t, jn) /* wave-uniform */
                                                 : (PairPark<Model, K>::need_t(t) ? PX[PairPark<Model, K>::row_t(t) * 64 + lane] : 0.0);
            int ab = (obs_fixed & nr.mask_at(jn)) ? -1 : 0;
            if (fiber_abs) ab = 1;
            int lo, hi;
            ab = vary_neighbors(jn, N, bck, ab, lo, hi, A.cends);
            int ui;
            FPP_STAMP(8) // exchange reads + stencil assembly + flags
            // FORCED (policy evaluation) is a separate instantiation: as a run-time flag it costs the minimising kernel 4 %
            int fu = -1;
            if constexpr (FORCED) fu = A.forced[(size_t)f * N + jn];
            const PairPre<Model, K> pre{PX, lane};
            const double val = node_backup<Model, FPP_CG, FPP_CGD, CandLds<Model>, true, PairPre<Model, K>>(A, ro, x, tv, cr, V, ab, ui, st, FORCED, fu, pre);
            FPP_STAMP(9) // control scan
            // lanes past the last fiber duplicate fiber F-1 and store the same numbers to the same place: no
            // divergent branch in the node loop (see node_backup on spilled lane tables)
            outv[(size_t)f * N + jn] = val;
            if (uidx) uidx[(size_t)f * N + jn] = ui;
            if (absorbed) absorbed[(size_t)f * N + jn] = ab;
        };

        // value of node N-2 (left neighbour of node 0 under a periodic boundary)
        double vwrap = 0.0;
        if (bck == C3SC_PERIODIC) {
            const double pv = partials(N - 2, [](int, double) __attribute__((always_inline)) {});
            if constexpr (H == 0) B0[NP * 64 + lane] = pv; // each wave writes the spare row of its own exchange block
            else B1[NP * 64 + lane] = pv;
            pair_barrier();
            vwrap = pv + (H == 0 ? B1[NP * 64 + lane] : B0[NP * 64 + lane]);
            pair_barrier();
        }
        double vone = 0.0;             // v[1]
        double v_m2 = 0.0, v_m1 = 0.0; // v[2t-2], v[2t-1]
        double Vd[NP];                 // wave 1: stencil of node 2t-1 waiting for v[2t]
#pragma unroll
        for (int g = 0; g < NP; g++) Vd[g] = 0.0;

        const int T = N / 2 + 1;
        for (int t = 0; t < T; t++) {
            const int j0 = 2 * t, j1 = 2 * t + 1;
            const bool has0 = j0 < N, has1 = j1 < N;
            double pv0 = 0.0, pv1 = 0.0;
            double Pown[NP]; // wave 0: its partial sums of the node it finalises stay in registers
#pragma unroll
            for (int g = 0; g < NP; g++) Pown[g] = 0.0;
            auto own_sink = [&](int g, double v) __attribute__((always_inline)) { Pown[g] = v; };
            if constexpr (K > 0 && K < D - 1 && FPP_PIPE2) {
                if (has0 && has1) { // both nodes: (c, a) of the second before the dots of the first
                    double chA[RH], ahA[RH], chB[RH], ahB[RH];
                    auto nofill = [](int) __attribute__((always_inline)) {};
                    auto value_of = [&](const double (&ch)[RH]) __attribute__((always_inline)) -> double {
                        double v = 0.0;
#pragma unroll
                        for (int i = 0; i < RH; i++) v = fma(sL[(H * RH + i) * 64 + lane], ch[i], v);
                        return v;
                    };
                    if constexpr (H == 0) {
                        auto sinkA = to_lds(B0);
                        ca_part(j1, chA, ahA, nofill);
                        ca_part(j0, chB, ahB, [&](int q) __attribute__((always_inline)) { dots_quarter(q, chA, ahA, sinkA); });
                        pv1 = value_of(chA);
                        sinkA(NV, pv1);
                        pv0 = dots_part(chB, ahB, own_sink);
                        B0[NP * 64 + lane] = pv0;
                    } else {
                        auto sinkA = to_lds(B1);
                        ca_part(j0, chA, ahA, nofill);
                        ca_part(j1, chB, ahB, [&](int q) __attribute__((always_inline)) { dots_quarter(q, chA, ahA, sinkA); });
                        pv0 = value_of(chA);
                        sinkA(NV, pv0);
                        pv1 = dots_part(chB, ahB, to_lds(B2));
                        B1[NP * 64 + lane] = pv1;
                    }
                } else if (has0) { // the last, unpaired node
                    if constexpr (H == 0) { pv0 = partials(j0, own_sink); B0[NP * 64 + lane] = pv0; }
                    else pv0 = partials(j0, to_lds(B1));
                }
            } else if constexpr (H == 0) { // the other wave's node first, the own node last
                if (has1) pv1 = partials(j1, to_lds(B0));
                if (has0) {
                    pv0 = partials(j0, own_sink);
                    B0[NP * 64 + lane] = pv0;
                }
            } else {
                if (has0) pv0 = partials(j0, to_lds(B1));
                if (has1) {
                    pv1 = partials(j1, to_lds(B2));
                    B1[NP * 64 + lane] = pv1;
                }
            }
            FPP_STAMP(3) // partials + LDS writes
            pair_barrier();
            FPP_STAMP(4) // barrier 1
            double v0 = 0.0, v1 = 0.0; // totals v[j0], v[j1]
            if constexpr (H == 0) {
                if (has1) v1 = pv1 + B1[NP * 64 + lane];
                if (has0) {
                    double P0[NP];
#pragma unroll
                    for (int g = 0; g < NP; g++) P0[g] = Pown[g] + B1[g * 64 + lane];
                    v0 = P0[NV];
                    double vlo, vhi;
                    dimk_values(j0, N, bck, v_m1, v0, v1, vwrap, (j0 == 0 ? v1 : vone), vlo, vhi);
                    finalize(j0, P0, vlo, vhi);
                }
            } else {
                if (has0) v0 = pv0 + B0[NP * 64 + lane];
                if (t >= 1) { // node 2t-1
                    const int jn = j0 - 1;
                    double vlo, vhi;
                    dimk_values(jn, N, bck, v_m2, Vd[NV], v0, vwrap, (jn == 1 ? Vd[NV] : vone), vlo, vhi);
                    finalize(jn, Vd, vlo, vhi);
                }
                if (has1) {
#pragma unroll
                    for (int g = 0; g < NP; g++) Vd[g] = B0[g * 64 + lane] + B2[g * 64 + lane];
                    v1 = Vd[NV];
                }
            }
            if (t == 0) vone = v1;
            v_m2 = v0;
            v_m1 = v1;
            FPP_STAMP(5) // LDS reads + finalize
            pair_barrier();
            FPP_STAMP(6) // barrier 2
        }
    }
    if (C3SC_STAMPS_ON && (A.dbg & 128) && lane == 0) {
        const size_t w = ((size_t)blockIdx.x * 2 + H) * 12;
        if (w < 65536 * 8)
            for (int i = 0; i < 12; i++) A.dbgbuf[w + i] = seg[i];
    }
}

// registers: two wavefronts per SIMD in general (256 VGPRs); the direct-fold kernels at ranks <= 6 sit at 122-137 and are asked to stay
// within 128 (four per SIMD: their LDS footprint allows it)
#ifndef FPP_WPS_LOW
#define FPP_WPS_LOW 4
#endif
#ifndef FPP_WPS_HIGH
#define FPP_WPS_HIGH 2 // wavefronts per SIMD of the staged (rank > 6) instantiations; 1 was measured (512 registers per wavefront, spills in AGPRs)
#endif
template <class Model, int RP>
__host__ __device__ constexpr int fpp_waves_per_simd() { return (fpp_direct<Model, RP>() && RP <= 6) ? FPP_WPS_LOW : (FPP_WPS_HIGH); }

template <class Model, int RP, int K, bool FORCED>
__global__ void __launch_bounds__(FPP_THREADS, (fpp_waves_per_simd<Model, RP>()))
    k_fiber_pair(const KArgs A, const double *__restrict__ ro, const int32_t *__restrict__ idx, double *__restrict__ outv,
                 int32_t *__restrict__ uidx, int32_t *__restrict__ absorbed)
{
    extern __shared__ double sKp[];
    unsigned st = 0;
    const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (h == 0) fiber_pair_body<Model, RP, K, 0, FORCED>(A, ro, idx, outv, uidx, absorbed, sKp, st);
    else fiber_pair_body<Model, RP, K, 1, FORCED>(A, ro, idx, outv, uidx, absorbed, sKp, st);
    if (st) atomicOr(A.status, st);
}

} // namespace c3sc
