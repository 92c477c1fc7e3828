// fold_lds.hpp -- folding helpers of the lane = fiber kernels (kernel_fiber_pair.hpp): a lane applies ITS fiber's matrix
// (G_m[i_m] with the lane's own i_m), read from a core staged once per tile in LDS with an odd node stride (kernel_fiber_pair.hpp: stage_core_image), to up to four
// vectors per pass; products are software-pipelined by hand (see the scheduling note below).  These were written for the
// one-wave-per-64-fibers kernel of round 1 (k_fiber_per_lane), which was retired in round 2: it was capped by the FP64
// issue rate of one wavefront per SIMD, AUTO never picked it, and it was the last kernel that depended on the "no
// lane-divergent branch may touch a lane-distributed table" rule.
#pragma once
#include <utility>

#include "kernel_common.hpp"

namespace c3sc {

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

// ---- the same products with the lane's matrix read straight from the cores in global memory (L2-resident): low-dimensional
// problems fold through at most d-2 matrices per fiber, and a staged copy of a whole core per 64 fibers costs more LDS (i.e.
// occupancy) than the gathered reads cost time
// v <- v G (ROWVEC: out[b] = sum_a v[a] G[a + b RP]) or v <- G v (out[a] = sum_b G[a + b RP] v[b]) for NV vectors, G read
// from global memory in chunks of whole storage columns (RP contiguous doubles each)
template <int RP, int NV, bool ROWVEC>
__device__ __forceinline__ void apply_glb(const double *__restrict__ G, double (&v)[NV][RP])
{
    // 16-18 doubles of the matrix in registers at a time.  Left alone the compiler hoists every load of the fold (three matrices
    // of a level, read through a const __restrict__ pointer: nothing orders them) above the first FMA and spills 200-500
    // registers.  An empty asm cannot fence them either -- it does not receive the noalias pointer, so loads through it may
    // cross it.  Instead the chunk's offset is laundered through a volatile asm (its loads depend on the asm's output)
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
        int off = b0 * RP; // (the OFFSET is laundered, not the pointer: a laundered pointer loses its address space and every load
        asm volatile("" : "+v"(off)); // through it becomes a FLAT load -- 90 of them in the dubins3d kernel until round 4)
        const double *Gc = G + off;
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

// neighbour values along the varying dim for node jn (nodeutil.c:570-624)
__device__ inline void dimk_values(int jn, int N, int bck, double vL, double vC, double vR, double vwrap, double vone,
                                   double &lo, double &hi)
{
    lo = vL;
    hi = vR;
    if (jn == 0) {
        lo = (bck == C3SC_PERIODIC) ? vwrap : vC;
        hi = (bck == C3SC_ABSORB) ? vC : vR;
    }
    if (jn == N - 1) {
        lo = (bck == C3SC_ABSORB) ? vC : vL;
        hi = (bck == C3SC_PERIODIC) ? vone : vC;
        if (N == 1) lo = vC;
    }
}

} // namespace c3sc
