// kernel_fiber_lane.hpp -- "one fiber per lane, one wavefront per workgroup" Bellman kernel for LOW ranks (gfx950).
//
// The fiber-pair kernel splits the rank index over two wavefronts because 2(d-1) folded vectors of r doubles do not fit
// 256 VGPRs at d = 7, r = 10.  At d <= 3, r <= 8 they are 64 VGPRs or fewer: here one wavefront owns 64 fibers outright.
// Same fold-once algebra (kernel_fiber_pair.hpp; valuef_eval_fiber_ind_nn src/valuefunc.c:369-585), but
//   * no partner, no LDS exchange rows, no barrier anywhere in the node loop: the wavefronts of a SIMD are independent
//     instruction streams, which is what an FP64 pipe with ~8-cycle dependent issue needs (the pair kernel's two waves
//     wait for each other twice per node pair, and its workgroup's staged core caps the occupancy at two waves per SIMD);
//   * no staging: a low-dimensional problem folds through at most d-2 matrices per fiber, so every lane reads the few
//     r x r matrices of ITS fiber straight from the cores (L2-resident, 38 KB for dubins3d) instead of the workgroup
//     copying a whole core into LDS per 64 fibers; LDS holds only the wave-uniform candidate and node tables;
//   * registers <= 128 (four wavefronts per SIMD) or <= 168 (three), chosen per instantiation.
// The varying core G_k[j] is the wave-uniform operand of the node loop (scalar loads, SGPR sources of v_fma_f64), the dim-k
// neighbours of a node are the previous / next node of the same lane (the backup of node j-1 waits for v_j in registers).
// Citations: process_fibers_neighbor src/nodeutil.c:489-627, bellman_optimal / bellman_control src/bellman.c:504-543,
// 367-480 (see kernel_common.hpp).
#pragma once
#include <utility>

#include "fold_lds.hpp"

#ifndef FL_CG
#define FL_CG 1 // candidates in flight in the discounted scan
#endif
#ifndef FL_CGD
#define FL_CGD 3 // ... and in the division-free (undiscounted) scan
#endif

namespace c3sc {

// v <- v G (ROWVEC: out[b] = sum_a v[a] G[a + b RP]) or v <- G v (out[a] = sum_b G[a + b RP] v[b]) for NV vectors, G read
// from global memory in chunks of whole storage columns (RP contiguous doubles each)
template <int RP, int NV, bool ROWVEC>
__device__ __forceinline__ void apply_glb(const double *__restrict__ G, double (&v)[NV][RP])
{
    // 16-18 doubles of the matrix in registers at a time.  Left alone the compiler hoists every load of the fold (three matrices
    // of a level, read through a const __restrict__ pointer: nothing orders them) above the first FMA and spills 200-500
    // registers.  An empty asm cannot fence them either -- it does not receive the noalias pointer, so loads through it may
    // cross it.  Instead the chunk's base pointer is laundered through a volatile asm (its loads depend on the asm's output)
    // and the accumulators are pinned by volatile asms after the chunk's FMAs: volatile asms keep their order, so chunk c+1 is
    // loaded after chunk c has been consumed.  The fold is short; other wavefronts cover its L2 round trips.
    constexpr int CH = (RP <= 4) ? RP : (RP <= 6 ? 3 : 2);
    double t[NV][RP];
#pragma unroll
    for (int s = 0; s < NV; s++)
#pragma unroll
        for (int i = 0; i < RP; i++) t[s][i] = 0.0;
#pragma unroll
    for (int b0 = 0; b0 < RP; b0 += CH) {
        const double *Gc = G + b0 * RP;
        asm volatile("" : "+v"(Gc));
        double g[CH][RP];
#pragma unroll
        for (int b = 0; b < CH; b++)
#pragma unroll
            for (int a = 0; a < RP; a++) g[b][a] = (b0 + b < RP) ? Gc[a + b * RP] : 0.0;
#pragma unroll
        for (int b = 0; b < CH; b++) {
            if (b0 + b < RP) {
#pragma unroll
                for (int s = 0; s < NV; s++)
#pragma unroll
                    for (int a = 0; a < RP; a++) {
                        if constexpr (ROWVEC) t[s][b0 + b] = fma(v[s][a], g[b][a], t[s][b0 + b]);
                        else t[s][a] = fma(g[b][a], v[s][b0 + b], t[s][a]);
                    }
            }
        }
#pragma unroll
        for (int s = 0; s < NV; s++)
#pragma unroll
            for (int i = 0; i < RP; i++) pin(t[s][i]);
    }
#pragma unroll
    for (int s = 0; s < NV; s++)
#pragma unroll
        for (int i = 0; i < RP; i++) v[s][i] = t[s][i];
}

// W[FIRST .. FIRST+COUNT) through G, at most two vectors per pass over the matrix
template <int RP, int NW, int FIRST, int COUNT, bool ROWVEC>
__device__ __forceinline__ void apply_range_glb(const double *__restrict__ G, double (&W)[NW][RP])
{
    if constexpr (COUNT > 0) {
        constexpr int NV = COUNT >= 2 ? 2 : 1;
        double tmp[NV][RP];
#pragma unroll
        for (int s = 0; s < NV; s++)
#pragma unroll
            for (int a = 0; a < RP; a++) tmp[s][a] = W[FIRST + s][a];
        apply_glb<RP, NV, ROWVEC>(G, tmp);
#pragma unroll
        for (int s = 0; s < NV; s++)
#pragma unroll
            for (int a = 0; a < RP; a++) W[FIRST + s][a] = tmp[s][a];
        apply_range_glb<RP, NW, FIRST + NV, COUNT - NV, ROWVEC>(G, W);
    }
}

template <class Model, int RP, int K, int WPS, bool FORCED>
__global__ void __launch_bounds__(64, WPS)
    k_fiber_lane(const KArgs A, const double *__restrict__ ro, const int32_t *__restrict__ idx, double *__restrict__ outv,
                 int32_t *__restrict__ uidx, int32_t *__restrict__ absorbed)
{
    constexpr int D = Model::D;
    constexpr int S = 2 * D + 1;
    constexpr int NV = 2 * (D - 1); // neighbour vectors, slot g = (m < K ? 2m : 2(m-1)) + s
    extern __shared__ double sTab[];
    const int lane = threadIdx.x;
    const int N = A.N;
    unsigned st = 0;
    const long ntiles = (A.F + 63) / 64;
    // wave-uniform candidate and node tables in LDS (read with broadcast addresses; nothing depends on inactive lanes)
    CandLds<Model> cr;
    NodeLds<Model, K> nr;
    {
        CandRegs<Model> cr0;
        cr0.load(A, ro);
        NodeRegs<Model, K> nr0;
        nr0.load(A, ro);
        cr.fill(sTab, cr0, A.ncand);
        nr.fill(sTab + CandLds<Model>::doubles(A.ncand), nr0, N);
        __syncthreads();
    }
    const int bck = A.bctype[K];
    const double *Gk = ro + A.core_off[K];

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long f_raw = tile * 64 + lane;
        const long f = f_raw < A.F ? f_raw : A.F - 1; // lanes past the end duplicate the last fiber (same stores, no divergence)
        // ---- the lane's fiber: fixed indices, their neighbours per boundary type (nodeutil.c:513-566)
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
        double tv[Model::NTAB > 0 ? Model::NTAB : 1];
        table_values<Model>(A, ro, fi, tv);

        double L[RP], R[RP], W[NV > 0 ? NV : 1][RP];
#pragma unroll
        for (int a = 0; a < RP; a++) { L[a] = (a == 0) ? 1.0 : 0.0; R[a] = (a == 0) ? 1.0 : 0.0; }

        // ---- prefix side: cores 0 .. K-1 into L and the left neighbour vectors
        if constexpr (K > 0) {
            {
                const double *c0 = ro + A.core_off[0];
#pragma unroll
                for (int b = 0; b < RP; b++) {
                    L[b] = c0[(size_t)fi[0] * RP + b];
                    W[0][b] = c0[(size_t)nbm[0] * RP + b];
                    W[1][b] = c0[(size_t)nbp[0] * RP + b];
                }
            }
            auto left_step = [&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                const double *cm = ro + A.core_off[m];
                const double *G = cm + (size_t)fi[m] * (RP * RP);
                apply_range_glb<RP, NV, 0, 2 * m, true>(G, W); // the vectors of the dims before m
                double t0[1][RP], t1[1][RP];
#pragma unroll
                for (int a = 0; a < RP; a++) { t0[0][a] = L[a]; t1[0][a] = L[a]; }
                apply_glb<RP, 1, true>(cm + (size_t)nbm[m] * (RP * RP), t0);
                apply_glb<RP, 1, true>(cm + (size_t)nbp[m] * (RP * RP), t1);
                double tl[1][RP];
#pragma unroll
                for (int a = 0; a < RP; a++) tl[0][a] = L[a];
                apply_glb<RP, 1, true>(G, tl);
#pragma unroll
                for (int a = 0; a < RP; a++) { W[2 * m][a] = t0[0][a]; W[2 * m + 1][a] = t1[0][a]; L[a] = tl[0][a]; }
            };
            [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (left_step(std::integral_constant<int, Ms + 1>{}), ...); }
            (std::make_integer_sequence<int, (K > 1 ? K - 1 : 0)>{});
        }
        // ---- suffix side: cores D-1 .. K+1 into R and the right neighbour vectors
        if constexpr (K < D - 1) {
            {
                const double *cl = ro + A.core_off[D - 1];
#pragma unroll
                for (int a = 0; a < RP; a++) {
                    R[a] = cl[(size_t)fi[D - 1] * RP + a];
                    W[2 * (D - 2)][a] = cl[(size_t)nbm[D - 1] * RP + a];
                    W[2 * (D - 2) + 1][a] = cl[(size_t)nbp[D - 1] * RP + a];
                }
            }
            auto right_step = [&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value; // D-2 down to K+1
                constexpr int g0 = 2 * (m - 1);        // this dim's pair; the dims above it are g0+2 .. NV-1
                const double *cm = ro + A.core_off[m];
                const double *G = cm + (size_t)fi[m] * (RP * RP);
                apply_range_glb<RP, NV, g0 + 2, NV - (g0 + 2), false>(G, W);
                double t0[1][RP], t1[1][RP], tr[1][RP];
#pragma unroll
                for (int a = 0; a < RP; a++) { t0[0][a] = R[a]; t1[0][a] = R[a]; tr[0][a] = R[a]; }
                apply_glb<RP, 1, false>(cm + (size_t)nbm[m] * (RP * RP), t0);
                apply_glb<RP, 1, false>(cm + (size_t)nbp[m] * (RP * RP), t1);
                apply_glb<RP, 1, false>(G, tr);
#pragma unroll
                for (int a = 0; a < RP; a++) { W[g0][a] = t0[0][a]; W[g0 + 1][a] = t1[0][a]; R[a] = tr[0][a]; }
            };
            [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (right_step(std::integral_constant<int, D - 2 - Ms>{}), ...); }
            (std::make_integer_sequence<int, (D - 2 - K > 0 ? D - 2 - K : 0)>{});
        }

        // ---- node loop: FT values of node j, then (one node later, when v_j is known) the Bellman backup of node j-1
        // c = G_k[j] R, a = L G_k[j], v_j = L c; the matrix element is wave-uniform (scalar load, SGPR operand)
        auto node_ft = [&](int j, double (&V)[S]) __attribute__((always_inline)) -> double {
            double c[RP], a[RP], vj;
            if constexpr (K == 0) { // 1 x r row
#pragma unroll
                for (int b = 0; b < RP; b++) { a[b] = Gk[(size_t)j * RP + b]; c[b] = 0.0; }
                vj = dot_reg<RP>(a, R);
            } else if constexpr (K == D - 1) { // r x 1 column
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
#pragma unroll
            for (int g = 0; g < NV; g++) {
                if (g < 2 * K) V[g] = dot_reg<RP>(W[g], c);
                else V[g + 2] = dot_reg<RP>(a, W[g]);
            }
            V[2 * K] = 0.0;
            V[2 * K + 1] = 0.0;
            V[2 * D] = vj;
            return vj;
        };
        double vwrap = 0.0; // v[N-2]: the left neighbour of node 0 under a periodic boundary
        if (bck == C3SC_PERIODIC && N > 1) {
            double Vt[S];
            vwrap = node_ft(N - 2 > 0 ? N - 2 : 0, Vt);
        }
        double vone = 0.0;            // v[1]
        double v_pp = 0.0, v_p = 0.0; // v[j-2], v[j-1]
        double Vp[S];                 // stencil of node j-1, waiting for v_j
#pragma unroll
        for (int s = 0; s < S; s++) Vp[s] = 0.0;

        for (int j = 0; j <= N; j++) {
            double V[S];
#pragma unroll
            for (int s = 0; s < S; s++) V[s] = 0.0;
            double vj = 0.0;
            if (j < N) { // wave-uniform
                vj = node_ft(j, V);
                if (j == 1) vone = vj;
            }
            if (j >= 1) {
                const int jn = j - 1;
                double vlo, vhi;
                dimk_values(jn, N, bck, v_pp, v_p, vj, vwrap, (jn == 0 ? vj : vone), vlo, vhi);
                Vp[2 * K] = vlo;
                Vp[2 * K + 1] = vhi;
                x[K] = nr.x_at(jn);
#pragma unroll
                for (int t = 0; t < Model::NTAB; t++)
                    if (Model::tab_dim(t) == K) tv[t] = nr.tab_at(t, jn); // wave-uniform
                int ab = (obs_fixed & nr.mask_at(jn)) ? -1 : 0;
                if (fiber_abs) ab = 1;
                int lo, hi;
                ab = vary_neighbors(jn, N, bck, ab, lo, hi, A.cends);
                int ui;
                int fu = -1;
                if constexpr (FORCED) fu = A.forced[(size_t)f * N + jn];
                const double val = node_backup<Model, FL_CG, FL_CGD, CandLds<Model>>(A, ro, x, tv, cr, Vp, ab, ui, st, FORCED, fu);
                outv[(size_t)f * N + jn] = val;
                if (uidx) uidx[(size_t)f * N + jn] = ui;
                if (absorbed) absorbed[(size_t)f * N + jn] = ab;
            }
#pragma unroll
            for (int s = 0; s < S; s++) Vp[s] = V[s];
            v_pp = v_p;
            v_p = vj;
        }
    }
    if (st) atomicOr(A.status, st);
}

} // namespace c3sc
