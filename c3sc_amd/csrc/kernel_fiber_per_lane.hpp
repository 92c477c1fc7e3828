// kernel_fiber_per_lane.hpp -- "one fiber per lane, node loop" Bellman kernel (gfx950, wave64).
//
// Why: in the fiber-per-wave kernel a wavefront owns ONE fiber, so 23 of 64 lanes idle at N = 41, the
// per-fiber prefix/suffix work runs on 30 lanes, and every node drags its r-vector through d-2 r x r
// cores.  Here a lane owns a whole fiber and the wavefront walks the nodes j = 0..N-1 together:
//   * the varying core G_k[j] is THE wave-uniform operand (same j for all 64 fibers) -> scalar loads,
//     v_fma_f64 with an SGPR matrix element, every lane busy for any N;
//   * everything that is constant along a fiber is folded ONCE into 2(d-1)+2 short vectors held in
//     registers:  L = G_0[i_0]..G_{k-1}[i_{k-1}],  R = G_{k+1}[i_{k+1}]..G_{d-1}[i_{d-1}],
//       w_m^{-+} = L_{m-1} G_m[i_m -+ 1] G_{m+1}[i_{m+1}] .. G_{k-1}[i_{k-1}]      (m < k, length r_k)
//       z_m^{-+} = G_{k+1}[i_{k+1}] .. G_{m-1}[i_{m-1}] G_m[i_m -+ 1] R_{m+1}      (m > k, length r_{k+1})
//     so that per node only  c = G_k[j] R,  a = L G_k[j]  (2 r^2 FMAs) and 2d-1 length-r dots remain:
//       V(j) = L.c,   V_m^{-+}(j) = w_m^{-+}.c  (m < k),   V_m^{-+}(j) = a.z_m^{-+}  (m > k).
//     Same numbers as valuef_eval_fiber_ind_nn (src/valuefunc.c:369-585), ~640 instead of ~1500 flop/node
//     at d = 7, r = 10;
//   * the per-fiber folding needs a DIFFERENT matrix per lane (G_m[i_m] with the lane's own i_m): each
//     fixed core is staged once per 256-fiber tile in LDS (odd node stride -> spread banks) and every
//     lane reads its own matrix from there, one LDS read feeding up to 4 FMAs (4 vectors per pass);
//   * the dim-k neighbours are the previous / next node of the same lane: the control minimisation of
//     node j-1 is delayed until V(j) is known (register pipeline, no LDS, handles reflect / periodic
//     ends with two saved values);
//   * 256-thread workgroups, one per CU, ~400 VGPRs (the vectors live in registers), grid-stride over
//     256-fiber tiles.
//
// Template parameter K = dim_vary is a compile-time constant so that all vector arrays are fully
// unrolled into registers.
#pragma once
#include <utility>

#include "kernel_common.hpp"

namespace c3sc {

constexpr int FPL_THREADS = 256;
#ifndef FPL_NV
#define FPL_NV 2
#endif
// keep the scheduler from hoisting whole matrices of loads ahead of their FMAs (register pressure)
#ifndef FPL_NO_FENCE
#define FPL_SCHED_FENCE() asm volatile("" ::: "memory")
#else
#define FPL_SCHED_FENCE()
#endif

__host__ __device__ constexpr int fpl_lds_stride(int elems) { return elems | 1; } // odd #doubles per node

// Scheduling control.  hipcc's schedulers happily hoist every LDS load of an unrolled r x r product above
// the FMAs (200 live VGPRs per matrix -> AGPR/scratch spills that made the first version of this kernel
// 20x slower than its instruction count).  The products below are therefore software-pipelined BY HAND:
// column b+1 is loaded while column b is consumed, and an empty asm that "modifies" the accumulators and
// clobbers memory pins each column's FMAs between its neighbours' loads.
__device__ __forceinline__ void pin(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void mem_fence() { asm volatile("" ::: "memory"); }

// out[b] = sum_a v[a] G[a + b*RP]   (row vector times the lane's own matrix in LDS), NV vectors per pass.
// Walks the matrix ROW by row so that the RP outputs are RP independent FMA chains (a column-wise walk is one
// RP-deep dependent chain per output and is latency-bound at two wavefronts per SIMD).
template <int RP, int NV>
__device__ inline void vecmat_lds(const double *G, double (&v)[NV][RP])
{
    double t[NV][RP];
    double g[2][RP];
#pragma unroll
    for (int s = 0; s < NV; s++)
#pragma unroll
        for (int b = 0; b < RP; b++) t[s][b] = 0.0;
#pragma unroll
    for (int b = 0; b < RP; b++) g[0][b] = G[b * RP];
#pragma unroll
    for (int a = 0; a < RP; a++) {
        if (a + 1 < RP) {
#pragma unroll
            for (int b = 0; b < RP; b++) g[(a + 1) & 1][b] = G[(a + 1) + b * RP];
        }
#pragma unroll
        for (int s = 0; s < NV; s++)
#pragma unroll
            for (int b = 0; b < RP; b++) {
                t[s][b] = fma(v[s][a], g[a & 1][b], t[s][b]);
                pin(t[s][b]);
            }
        mem_fence();
    }
#pragma unroll
    for (int s = 0; s < NV; s++)
#pragma unroll
        for (int b = 0; b < RP; b++) v[s][b] = t[s][b];
}

// out[a] = sum_b G[a + b*RP] v[b]   (matrix times column vector)
template <int RP, int NV>
__device__ inline void matvec_lds(const double *G, double (&v)[NV][RP])
{
    double t[NV][RP];
    double g[2][RP];
#pragma unroll
    for (int s = 0; s < NV; s++)
#pragma unroll
        for (int a = 0; a < RP; a++) t[s][a] = 0.0;
#pragma unroll
    for (int a = 0; a < RP; a++) g[0][a] = G[a];
#pragma unroll
    for (int b = 0; b < RP; b++) {
        if (b + 1 < RP) {
#pragma unroll
            for (int a = 0; a < RP; a++) g[(b + 1) & 1][a] = G[a + (b + 1) * RP];
        }
#pragma unroll
        for (int s = 0; s < NV; s++)
#pragma unroll
            for (int a = 0; a < RP; a++) {
                t[s][a] = fma(g[b & 1][a], v[s][b], t[s][a]);
                pin(t[s][a]);
            }
        mem_fence();
    }
#pragma unroll
    for (int s = 0; s < NV; s++)
#pragma unroll
        for (int a = 0; a < RP; a++) v[s][a] = t[s][a];
}

// apply `op` to W[FIRST .. FIRST+COUNT) in passes of at most 4 vectors (compile-time recursion)
template <int RP, int NW, int FIRST, int COUNT, bool ROWVEC, int NVMAX = FPL_NV>
__device__ inline void apply_core(const double *G, double (&W)[NW][RP])
{
    if constexpr (COUNT > 0) {
        constexpr int NV = COUNT >= NVMAX ? NVMAX : COUNT;
        double tmp[NV][RP];
#pragma unroll
        for (int s = 0; s < NV; s++)
#pragma unroll
            for (int a = 0; a < RP; a++) tmp[s][a] = W[FIRST + s][a];
        if constexpr (ROWVEC) vecmat_lds<RP, NV>(G, tmp);
        else matvec_lds<RP, NV>(G, tmp);
#pragma unroll
        for (int s = 0; s < NV; s++)
#pragma unroll
            for (int a = 0; a < RP; a++) W[FIRST + s][a] = tmp[s][a];
        apply_core<RP, NW, FIRST + NV, COUNT - NV, ROWVEC, NVMAX>(G, W);
    }
}

template <int RP>
__device__ inline double dot_reg(const double (&a)[RP], const double (&b)[RP])
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < RP; i++) s = fma(a[i], b[i], s);
    return s;
}

// cooperative global -> LDS copy of one core with the padded node stride
__device__ inline void stage_core_n(double *sK, const double *__restrict__ src, int n_nodes, int elems, int stride, int nthreads)
{
    // The copy is pure latency (33 KB from L2 per core and tile): every thread moves PAIRS of doubles (16-byte
    // loads; elems is even, so a pair never straddles two nodes) and keeps a batch of 8 loads in flight before
    // the first LDS write.  dst index of element e = e + (e / elems) * (stride - elems).  Trip counts are the same
    // for every thread; past the end a thread repeats the last pair (same values to the same slots): no
    // lane-divergent branch or loop exit.
    const int pairs = (n_nodes * elems) >> 1, pad = stride - elems;
    constexpr int B = 8;
    const int per_batch = B * nthreads;
    for (int base = 0; base < pairs; base += per_batch) {
        double2 buf[B];
        int pe[B];
#pragma unroll
        for (int q = 0; q < B; q++) {
            const int p = base + q * nthreads + (int)threadIdx.x;
            pe[q] = 2 * (p < pairs ? p : pairs - 1);
            buf[q] = *reinterpret_cast<const double2 *>(src + pe[q]);
        }
#pragma unroll
        for (int q = 0; q < B; q++) {
            const int node = pe[q] / elems; // elems is a small wave-uniform value: one multiply-high
            double *d = sK + pe[q] + node * pad;
            d[0] = buf[q].x;
            d[1] = buf[q].y;
        }
    }
}

__device__ inline void stage_core(double *sK, const double *__restrict__ src, int n_nodes, int elems, int stride)
{
    stage_core_n(sK, src, n_nodes, elems, stride, FPL_THREADS);
}

// number of neighbour vectors that live in LDS instead of registers: the 2(D-1) vectors need
// 2(D-1)*RP*2 VGPRs, but only 256 VGPRs are addressable by VALU instructions (the other 256 of the
// unified file are AGPRs).  Vectors g < NLDS sit in LDS as [g][c/2][lane][2] (one conflict-free
// ds_read_b128 per pair of components), the rest in registers.
template <int D, int RP>
__host__ __device__ constexpr int fpl_nlds()
{
    const int nv = 2 * (D - 1);
    const int max_reg_vecs = 120 / (2 * RP) > 0 ? 120 / (2 * RP) : 0; // ~120 VGPRs for resident vectors
    return nv > max_reg_vecs ? nv - max_reg_vecs : 0;
}

template <class Model, int RP, int K>
__global__ void __launch_bounds__(FPL_THREADS, 1)
    k_fiber_per_lane(const KArgs A, const double *__restrict__ ro, const int32_t *__restrict__ idx, double *__restrict__ outv,
                     int32_t *__restrict__ uidx, int32_t *__restrict__ absorbed)
{
    constexpr int D = Model::D;
    constexpr int S = 2 * D + 1;
    constexpr int NV = 2 * (D - 1);   // neighbour vectors, index g = (m < K ? 2m : 2(m-1)) + s
    constexpr int NLDS = fpl_nlds<D, RP>();
    constexpr int NREG = NV - NLDS;
    constexpr int NREGa = NREG > 0 ? NREG : 1;
    constexpr int RPe = RP + (RP & 1); // components per LDS vector, even
    static_assert(RP % 2 == 0 || NLDS == 0, "LDS-resident vectors need an even padded rank");
    extern __shared__ double smem[];
    double *sW = smem;                                      // [NLDS][RPe/2][256][2]
    double *sK = smem + (size_t)NLDS * RPe * FPL_THREADS;   // staged core
    const int N = A.N;
    const int tid = threadIdx.x;
    unsigned st = 0;
    const long ntiles = (A.F + FPL_THREADS - 1) / FPL_THREADS;
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#define FPL_STAMP(slot)                                                   \
    if (C3SC_STAMPS_ON && (A.dbg & 128)) {                                \
        const unsigned long long now__ = clock64();                       \
        seg[slot] += now__ - tlast;                                       \
        tlast = now__;                                                    \
    }
    CandRegs<Model> cr;
    cr.load(A, ro);
    NodeRegs<Model, K> nr;
    nr.load(A, ro);

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long f_raw = tile * FPL_THREADS + threadIdx.x;
        const bool live = f_raw < A.F;
        const long f = live ? f_raw : A.F - 1;

        // ---- the lane's fiber: fixed indices, neighbour indices per boundary type (nodeutil.c:513-566)
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

        double L[RP], R[RP], WR[NREGa][RP];
#pragma unroll
        for (int a = 0; a < RP; a++) { L[a] = (a == 0) ? 1.0 : 0.0; R[a] = (a == 0) ? 1.0 : 0.0; }

        // vector g: load / store (LDS slot of this lane or register array)
        auto vload = [&](auto gc, double (&v)[RP]) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < NLDS) {
#pragma unroll
                for (int c = 0; c < RP; c++) v[c] = sW[(((size_t)g * (RPe / 2) + c / 2) * FPL_THREADS + tid) * 2 + (c & 1)];
            } else {
#pragma unroll
                for (int c = 0; c < RP; c++) v[c] = WR[g - NLDS][c];
            }
        };
        auto vstore = [&](auto gc, const double (&v)[RP]) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < NLDS) {
#pragma unroll
                for (int c = 0; c < RP; c++) sW[(((size_t)g * (RPe / 2) + c / 2) * FPL_THREADS + tid) * 2 + (c & 1)] = v[c];
            } else {
#pragma unroll
                for (int c = 0; c < RP; c++) WR[g - NLDS][c] = v[c];
            }
        };

        FPL_STAMP(0) // tile setup
        // ---- prefix side: fold cores 0..K-1 into L and the left neighbour vectors (g = 2m + s)
        if constexpr (K > 0) {
            { // core 0 is a 1 x r row per node
                constexpr int str = fpl_lds_stride(RP);
                __syncthreads();
                stage_core(sK, ro + A.core_off[0], A.ngrid[0], RP, str);
                __syncthreads();
                double q0[RP], q1[RP];
#pragma unroll
                for (int b = 0; b < RP; b++) {
                    L[b] = sK[fi[0] * str + b];
                    q0[b] = sK[nbm[0] * str + b];
                    q1[b] = sK[nbp[0] * str + b];
                }
                vstore(std::integral_constant<int, 0>{}, q0);
                vstore(std::integral_constant<int, 1>{}, q1);
            }
            auto left_step = [&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                constexpr int str = fpl_lds_stride(RP * RP);
                __syncthreads();
                stage_core(sK, ro + A.core_off[m], A.ngrid[m], RP * RP, str);
                __syncthreads();
                const double *G = sK + fi[m] * str;
                // push the existing left vectors through G_m[i_m], one at a time
                [&]<int... Gs>(std::integer_sequence<int, Gs...>) {
                    (([&] {
                         double t[1][RP];
                         vload(std::integral_constant<int, Gs>{}, t[0]);
                         vecmat_lds<RP, 1>(G, t);
                         vstore(std::integral_constant<int, Gs>{}, t[0]);
                     }()),
                     ...);
                }(std::make_integer_sequence<int, 2 * m>{});
                // the new pair from the running prefix
                {
                    double t0[1][RP], t1[1][RP];
#pragma unroll
                    for (int a = 0; a < RP; a++) { t0[0][a] = L[a]; t1[0][a] = L[a]; }
                    vecmat_lds<RP, 1>(sK + nbm[m] * str, t0);
                    vstore(std::integral_constant<int, 2 * m>{}, t0[0]);
                    vecmat_lds<RP, 1>(sK + nbp[m] * str, t1);
                    vstore(std::integral_constant<int, 2 * m + 1>{}, t1[0]);
                }
                {
                    double t[1][RP];
#pragma unroll
                    for (int a = 0; a < RP; a++) t[0][a] = L[a];
                    vecmat_lds<RP, 1>(G, t);
#pragma unroll
                    for (int a = 0; a < RP; a++) L[a] = t[0][a];
                }
            };
            [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (left_step(std::integral_constant<int, Ms + 1>{}), ...); }
            (std::make_integer_sequence<int, (K > 1 ? K - 1 : 0)>{});
        }

        // ---- suffix side: fold cores D-1..K+1 into R and the right neighbour vectors (g = 2(m-1) + s)
        if constexpr (K < D - 1) {
            { // core D-1 is an r x 1 column per node
                constexpr int str = fpl_lds_stride(RP);
                __syncthreads();
                stage_core(sK, ro + A.core_off[D - 1], A.ngrid[D - 1], RP, str);
                __syncthreads();
                double q0[RP], q1[RP];
#pragma unroll
                for (int a = 0; a < RP; a++) {
                    R[a] = sK[fi[D - 1] * str + a];
                    q0[a] = sK[nbm[D - 1] * str + a];
                    q1[a] = sK[nbp[D - 1] * str + a];
                }
                vstore(std::integral_constant<int, 2 * (D - 2)>{}, q0);
                vstore(std::integral_constant<int, 2 * (D - 2) + 1>{}, q1);
            }
            auto right_step = [&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value; // D-2 down to K+1
                constexpr int str = fpl_lds_stride(RP * RP);
                constexpr int g0 = 2 * (m - 1);        // this dim's pair; the dims above it are g0+2 .. NV-1
                __syncthreads();
                stage_core(sK, ro + A.core_off[m], A.ngrid[m], RP * RP, str);
                __syncthreads();
                const double *G = sK + fi[m] * str;
                [&]<int... Gs>(std::integer_sequence<int, Gs...>) {
                    (([&] {
                         double t[1][RP];
                         vload(std::integral_constant<int, g0 + 2 + Gs>{}, t[0]);
                         matvec_lds<RP, 1>(G, t);
                         vstore(std::integral_constant<int, g0 + 2 + Gs>{}, t[0]);
                     }()),
                     ...);
                }(std::make_integer_sequence<int, NV - (g0 + 2)>{});
                {
                    double t0[1][RP], t1[1][RP];
#pragma unroll
                    for (int a = 0; a < RP; a++) { t0[0][a] = R[a]; t1[0][a] = R[a]; }
                    matvec_lds<RP, 1>(sK + nbm[m] * str, t0);
                    vstore(std::integral_constant<int, g0>{}, t0[0]);
                    matvec_lds<RP, 1>(sK + nbp[m] * str, t1);
                    vstore(std::integral_constant<int, g0 + 1>{}, t1[0]);
                }
                {
                    double t[1][RP];
#pragma unroll
                    for (int a = 0; a < RP; a++) t[0][a] = R[a];
                    matvec_lds<RP, 1>(G, t);
#pragma unroll
                    for (int a = 0; a < RP; a++) R[a] = t[0][a];
                }
            };
            [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (right_step(std::integral_constant<int, D - 2 - Ms>{}), ...); }
            (std::make_integer_sequence<int, (D - 2 - K > 0 ? D - 2 - K : 0)>{});
        }

        FPL_STAMP(1) // folding
        // ---- node loop: FT values of node j, then (delayed by one node) the Bellman backup of node j-1
        const int bck = A.bctype[K];
        const double *Gk = ro + A.core_off[K];
        // dot of neighbour vector g with a register vector
        auto vdot = [&](auto gc, const double (&u)[RP]) __attribute__((always_inline)) -> double {
            constexpr int g = decltype(gc)::value;
            double acc0 = 0.0, acc1 = 0.0; // two chains per dot: halves the dependent-FMA latency
            if constexpr (g < NLDS) {
#pragma unroll
                for (int c = 0; c < RP; c += 2) {
                    acc0 = fma(sW[(((size_t)g * (RPe / 2) + c / 2) * FPL_THREADS + tid) * 2], u[c], acc0);
                    if (c + 1 < RP) acc1 = fma(sW[(((size_t)g * (RPe / 2) + c / 2) * FPL_THREADS + tid) * 2 + 1], u[c + 1], acc1);
                }
                double acc = acc0 + acc1;
                pin(acc);
                if constexpr (g % 3 == 2 || g == NLDS - 1) mem_fence(); // at most 3 vectors of LDS loads in flight
                return acc;
            } else {
#pragma unroll
                for (int c = 0; c < RP; c += 2) {
                    acc0 = fma(WR[g - NLDS][c], u[c], acc0);
                    if (c + 1 < RP) acc1 = fma(WR[g - NLDS][c + 1], u[c + 1], acc1);
                }
                return acc0 + acc1;
            }
        };
        // value of one node only (needed ahead of time for the periodic wrap: lo(0) = N-2)
        auto node_value = [&](int j) __attribute__((always_inline)) -> double {
            if constexpr (K == 0) {
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < RP; b++) v = fma(Gk[(size_t)j * RP + b], R[b], v);
                return v;
            } else if constexpr (K == D - 1) {
                double v = 0.0;
#pragma unroll
                for (int a = 0; a < RP; a++) v = fma(L[a], Gk[(size_t)j * RP + a], v);
                return v;
            } else {
                const double *G = Gk + (size_t)j * RP * RP;
                double v = 0.0;
#pragma unroll
                for (int a = 0; a < RP; a++) {
                    double ca = 0.0;
#pragma unroll
                    for (int b = 0; b < RP; b++) ca = fma(G[a + b * RP], R[b], ca);
                    v = fma(L[a], ca, v);
                }
                return v;
            }
        };
        double v_wrap_lo = 0.0; // v[N-2], the left neighbour of node 0 under a periodic boundary (quirk Q8)
        if (bck == C3SC_PERIODIC) v_wrap_lo = node_value(N - 2);
        double v_one = 0.0;     // v[1], the right neighbour of node N-1 under a periodic boundary

        double Vp[S];           // stencil of the previous node, waiting for its right neighbour
        double v_pp = 0.0, v_p = 0.0; // v[j-2], v[j-1]
#pragma unroll
        for (int s = 0; s < S; s++) Vp[s] = 0.0;

        for (int j = 0; j <= N; j++) {
            double V[S];
            double vj = 0.0;
            if (j < N) {
                double c[RP], a[RP];
                if constexpr (K == 0) {
#pragma unroll
                    for (int b = 0; b < RP; b++) { a[b] = Gk[(size_t)j * RP + b]; c[b] = 0.0; }
                    vj = dot_reg<RP>(a, R);
                } else if constexpr (K == D - 1) {
#pragma unroll
                    for (int i = 0; i < RP; i++) { c[i] = Gk[(size_t)j * RP + i]; a[i] = 0.0; }
                    vj = dot_reg<RP>(L, c);
                } else {
                    const double *G = Gk + (size_t)j * RP * RP;
#pragma unroll
                    for (int i = 0; i < RP; i++) c[i] = 0.0;
#pragma unroll
                    for (int b = 0; b < RP; b++) {
                        double s = 0.0;
#pragma unroll
                        for (int i = 0; i < RP; i++) {
                            const double g = G[i + b * RP];
                            c[i] = fma(g, R[b], c[i]);
                            s = fma(L[i], g, s);
                        }
                        a[b] = s;
                    }
                    vj = dot_reg<RP>(L, c);
                }
                // neighbour values: left vectors against c, right vectors against a
                [&]<int... Gs>(std::integer_sequence<int, Gs...>) {
                    ((V[(Gs < 2 * K) ? Gs : Gs + 2] = (Gs < 2 * K) ? vdot(std::integral_constant<int, Gs>{}, c)
                                                                   : vdot(std::integral_constant<int, Gs>{}, a)),
                     ...);
                }(std::make_integer_sequence<int, NV>{});
                V[2 * D] = vj;
                V[2 * K] = 0.0;
                V[2 * K + 1] = 0.0;
                if (j == 1) v_one = vj;
            }
            FPL_STAMP(2) // FT of node j
            if (j >= 1) { // node j-1 now knows both neighbours along dim K (nodeutil.c:570-624)
                const int jn = j - 1;
                double vlo = v_pp, vhi = vj;
                if (jn == 0) {
                    vlo = (bck == C3SC_PERIODIC) ? v_wrap_lo : v_p;
                    vhi = (bck == C3SC_ABSORB) ? v_p : vj;
                }
                if (jn == N - 1) {
                    vlo = (bck == C3SC_ABSORB) ? v_p : v_pp;
                    vhi = (bck == C3SC_PERIODIC) ? v_one : v_p;
                }
                Vp[2 * K] = vlo;
                Vp[2 * K + 1] = vhi;
                // node coordinates, obstacle / face / end-point flags (nodeutil.c:496-624)
                x[K] = nr.x_at(jn);
#pragma unroll
                for (int t = 0; t < Model::NTAB; t++)
                    if (Model::tab_dim(t) == K) tv[t] = nr.tab_at(t, jn); // wave-uniform
                int ab = (obs_fixed & nr.mask_at(jn)) ? -1 : 0;
                if (fiber_abs) ab = 1;
                int lo, hi;
                ab = vary_neighbors(jn, N, bck, ab, lo, hi);
                int ui;
                const bool forced = A.forced != nullptr; // wave-uniform
                const int fu = forced ? A.forced[(size_t)f * N + jn] : -1;
                const double val = node_backup<Model, 3>(A, ro, x, tv, cr, Vp, ab, ui, st, forced, fu);
                // lanes past the last fiber duplicate fiber F-1: same numbers to the same place, no divergent branch
                outv[(size_t)f * N + jn] = val;
                if (uidx) uidx[(size_t)f * N + jn] = ui;
                if (absorbed) absorbed[(size_t)f * N + jn] = ab;
            }
            FPL_STAMP(3) // backup of node j-1
            if (j < N) {
#pragma unroll
                for (int s = 0; s < S; s++) Vp[s] = V[s];
                v_pp = v_p;
                v_p = vj;
            }
        }
    }
    if (C3SC_STAMPS_ON && (A.dbg & 128) && (threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
        if (w < 65536 * 8)
            for (int i = 0; i < 8; i++) A.dbgbuf[w + i] = seg[i];
    }
    if (st) atomicOr(A.status, st);
}

} // namespace c3sc
