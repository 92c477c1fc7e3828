// kernel_fiber_quad.hpp -- "sixteen fibers per wavefront, rank split over four lane groups" Bellman kernel (gfx950).
//
// Same fold-once algebra as kernel_fiber_pair.hpp (everything constant along a fiber is folded into L, R and the
// 2(d-1) neighbour vectors; per node only c = G_k[j] R, a = L G_k[j] and 2d-1 short dots remain), laid out for the
// matrix cores:
//
//   * lane l = (q, t) = (l / 16, l % 16): t is one of the wavefront's 16 fibers, q one of four rank quarters.  Lane
//     (q, t) keeps components [qC, (q+1)C), C = RP/4, of every folded vector of fiber t -- the per-fiber state
//     (2(d-1)+2 vectors of RP doubles) is spread over four lanes, so it fits the register file at any compiled rank
//     without a second wavefront, LDS exchange rows or barriers in the node loop.
//   * c = G_k[j] R and a = L G_k[j] share their matrix between all fibers: they run on the matrix cores as
//     v_mfma_f64_16x16x4_f64 with M = components (permuted so that the result lands in the owner lane), N = the 16
//     fibers, K = RP in C steps.  The A operand is a pre-permuted copy of the core (k_quad_aux) read with one coalesced
//     512-byte load per step; the B operand is the lane's own component of R / L; the four D registers of lane (q, t)
//     are exactly its components of c / a.
//   * the 2d-1 dots of a node are C FMAs per lane plus a sum over q.  Four nodes are processed together and the sum
//     over q is a transposing reduction in registers (v_permlane32_swap, v_permlane16_swap + 3 adds per four sums):
//     afterwards lane (q, t) holds the complete stencil of node j0+q of fiber t, and the control minimisation
//     (node_backup) runs with one (fiber, node) per lane -- all 64 lanes busy for any N.
//   * folding is level-synchronous over the workgroup: for each fixed dimension the WHOLE core is staged once in LDS
//     (coalesced; 51 KB at N = 25, rank 16) and every lane reads its fiber's matrix rows from there, so the cores cross
//     L2 -> CU once per tile of 16 x (waves per workgroup) fibers instead of three gathered matrices per fiber and
//     dimension.  A matrix-vector product is 4 C^2 FMAs per lane and a transposing reduction per output component; up to
//     four vectors share one pass over the matrix.  The level loop is a run-time loop (vector slots shift by one pair
//     per level), so the code size does not grow with d^2.
//
// Citations: valuef_eval_fiber_ind_nn src/valuefunc.c:369-585, process_fibers_neighbor src/nodeutil.c:489-627,
// bellman_optimal / bellman_control src/bellman.c:504-543, 367-480 (see kernel_common.hpp).
#pragma once
#include <type_traits>

#include "kernel_common.hpp"

#ifndef FQ_NVB
#define FQ_NVB 4 // vectors per pass over a staged matrix in the folding phase
#endif

namespace c3sc {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int quad_c(int rp) { return rp / 4; }            // components per lane
__host__ __device__ constexpr int quad_mb(int rp) { return (rp / 4 + 3) / 4; } // 16-row blocks of the MFMA M dimension
// doubles of one node's pre-permuted A operands: [product c / a][MB][C steps][64 lanes]
__host__ __device__ constexpr int quad_aop_node(int rp) { return 2 * quad_mb(rp) * quad_c(rp) * 64; }

// x (lanes 32-63) <-> y (lanes 0-31), both dwords of a double
__device__ __forceinline__ void swap32_f64(double &x, double &y)
{
    const unsigned long long bx = (unsigned long long)__double_as_longlong(x), by = (unsigned long long)__double_as_longlong(y);
    const v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)bx, (unsigned)by, false, false);
    const v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)(bx >> 32), (unsigned)(by >> 32), false, false);
    x = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
    y = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
// odd 16-lane rows of x <-> even rows of y
__device__ __forceinline__ void swap16_f64(double &x, double &y)
{
    const unsigned long long bx = (unsigned long long)__double_as_longlong(x), by = (unsigned long long)__double_as_longlong(y);
    const v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)bx, (unsigned)by, false, false);
    const v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)(bx >> 32), (unsigned)(by >> 32), false, false);
    x = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
    y = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}

// Transposing sum over the four lane groups: every lane passes its partial values of four items; lane (q, t) gets
// sum over q' of item q of lane (q', t).  Fixed association ((q, q^2) pairs first), identical on every run.
__device__ __forceinline__ double reduce4(double p0, double p1, double p2, double p3)
{
    swap32_f64(p0, p2); // lanes 0-31: p0 own, p2 = partner's p0;  lanes 32-63: p0 = partner's p2, p2 own
    swap32_f64(p1, p3);
    double s0 = p0 + p2, s1 = p1 + p3; // lanes 0-31: items 0, 1;  lanes 32-63: items 2, 3
    swap16_f64(s0, s1);
    return s0 + s1;
}

// v <- M v for the vectors X[B0 .. B0+NVB): M is this lane's fiber's matrix in LDS, row-major [x][y] with row stride RP
// (x = output component, y = input component; the left side stages the transposed core, so both sides are this product).
// Lane (q, t) holds y in [qC, (q+1)C) of every vector and ends with x in the same range.
// C consecutive doubles of a staged matrix row.  For even C the address is 16-byte aligned (node stride, row stride and q C
// are all even numbers of doubles), which the compiler cannot see: told so, it emits ds_read_b128 (256 B/clk) instead of
// ds_read2_b64 (128 B/clk) -- LDS is the busiest unit of the folding phase.
template <int C>
__device__ __forceinline__ void load_row(const double *p, double (&m)[C])
{
    if constexpr (C % 2 == 0) {
        const double2 *p2 = reinterpret_cast<const double2 *>(__builtin_assume_aligned(p, 16));
#pragma unroll
        for (int i = 0; i < C / 2; i++) { const double2 v = p2[i]; m[2 * i] = v.x; m[2 * i + 1] = v.y; }
    } else {
#pragma unroll
        for (int i = 0; i < C; i++) m[i] = p[i];
    }
}

template <int RP, int NX, int B0, int NVB>
__device__ __forceinline__ void apply_quad(const double *Mn, double (&X)[NX][RP / 4], int q)
{
    constexpr int C = RP / 4;
    double o[NVB][C];
    // The four rows of output group g are four independent FMA chains per vector, advanced together (a chain of its own per
    // row and vector is C dependent f64 FMAs with nothing in between: latency-bound at one or two wavefronts per SIMD), and the
    // rows of group g+1 are fetched from LDS while group g is summed.
    double mrow[2][4][C];
#pragma unroll
    for (int qq = 0; qq < 4; qq++) load_row<C>(Mn + (qq * C) * RP + q * C, mrow[0][qq]);
#pragma unroll
    for (int g = 0; g < C; g++) {
        if (g + 1 < C) {
#pragma unroll
            for (int qq = 0; qq < 4; qq++) load_row<C>(Mn + (qq * C + g + 1) * RP + q * C, mrow[(g + 1) & 1][qq]);
        }
        double P[NVB][4];
#pragma unroll
        for (int s = 0; s < NVB; s++)
#pragma unroll
            for (int qq = 0; qq < 4; qq++) P[s][qq] = mrow[g & 1][qq][0] * X[B0 + s][0];
#pragma unroll
        for (int i = 1; i < C; i++)
#pragma unroll
            for (int s = 0; s < NVB; s++)
#pragma unroll
                for (int qq = 0; qq < 4; qq++) P[s][qq] = fma(mrow[g & 1][qq][i], X[B0 + s][i], P[s][qq]);
#pragma unroll
        for (int s = 0; s < NVB; s++) o[s][g] = reduce4(P[s][0], P[s][1], P[s][2], P[s][3]);
    }
#pragma unroll
    for (int s = 0; s < NVB; s++)
#pragma unroll
        for (int g = 0; g < C; g++) X[B0 + s][g] = o[s][g];
}

// The two neighbour matrices of a level applied to the same vector v (the running prefix / suffix): the two products are
// interleaved, eight independent chains per output group.
template <int RP>
__device__ __forceinline__ void apply_quad_pair(const double *Mlo, const double *Mhi, const double (&v)[RP / 4], double (&olo)[RP / 4],
                                                double (&ohi)[RP / 4], int q)
{
    constexpr int C = RP / 4;
#pragma unroll
    for (int g = 0; g < C; g++) {
        double ml[4][C], mh[4][C];
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
            load_row<C>(Mlo + (qq * C + g) * RP + q * C, ml[qq]);
            load_row<C>(Mhi + (qq * C + g) * RP + q * C, mh[qq]);
        }
        double Pl[4], Ph[4];
#pragma unroll
        for (int qq = 0; qq < 4; qq++) { Pl[qq] = ml[qq][0] * v[0]; Ph[qq] = mh[qq][0] * v[0]; }
#pragma unroll
        for (int i = 1; i < C; i++)
#pragma unroll
            for (int qq = 0; qq < 4; qq++) { Pl[qq] = fma(ml[qq][i], v[i], Pl[qq]); Ph[qq] = fma(mh[qq][i], v[i], Ph[qq]); }
        olo[g] = reduce4(Pl[0], Pl[1], Pl[2], Pl[3]);
        ohi[g] = reduce4(Ph[0], Ph[1], Ph[2], Ph[3]);
    }
}

// one vector, result returned separately (the two neighbour matrices of a level are applied to the running prefix/suffix)
template <int RP>
__device__ __forceinline__ void apply_quad1(const double *Mn, const double (&v)[RP / 4], double (&out)[RP / 4], int q)
{
    double X[1][RP / 4];
#pragma unroll
    for (int i = 0; i < RP / 4; i++) X[0][i] = v[i];
    apply_quad<RP, 1, 0, 1>(Mn, X, q);
#pragma unroll
    for (int i = 0; i < RP / 4; i++) out[i] = X[0][i];
}

// all live vectors of one side through the matrix of this level, up to four per pass; blocks past the live count are skipped
template <int RP, int NX, int B0, int NVBMAX = FQ_NVB>
__device__ __forceinline__ void apply_live(const double *Mn, double (&X)[NX][RP / 4], int nlive, int q)
{
    if constexpr (B0 < NX) {
        constexpr int NVB = (NX - B0) < NVBMAX ? (NX - B0) : NVBMAX;
        if (B0 < nlive) apply_quad<RP, NX, B0, NVB>(Mn, X, q); // wave-uniform
        apply_live<RP, NX, B0 + NVB, NVBMAX>(Mn, X, nlive, q);
    }
}

// LDS stride of one node's matrix / vector: two doubles of padding shift consecutive nodes by four banks, so the 16 fibers
// of a wavefront (16 different nodes in general) do not all read the same bank (an unpadded rank^2 stride is a multiple of
// the 64 banks: a 16-way conflict on every read)
__host__ __device__ constexpr int quad_stride(int elems) { return elems + 2; }

// coalesced copy of n_nodes x elems doubles (elems even) from global memory into LDS rows of quad_stride(elems), whole workgroup
__device__ inline void stage_padded(double *dst, const double *__restrict__ src, int n_nodes, int elems)
{
    const int pairs = (n_nodes * elems) >> 1;
    for (int p = threadIdx.x; p < pairs; p += blockDim.x) {
        const int e = 2 * p, node = e / elems;
        *reinterpret_cast<double2 *>(dst + e + 2 * node) = *reinterpret_cast<const double2 *>(src + e);
    }
}

// c or a of one node on the matrix cores: out[g] = sum_y M[qC+g][y] v[y] with M = G_k[j] (c) or its transpose (a)
template <int RP>
__device__ __forceinline__ void mfma_prod(const double *__restrict__ aop /* [MB][C][64] */, const double (&v)[RP / 4],
                                          double (&out)[RP / 4], int lane)
{
    constexpr int C = RP / 4, MB = quad_mb(RP);
    v4d acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) acc[mb] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < C; s++)
#pragma unroll
        for (int mb = 0; mb < MB; mb++) acc[mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[(mb * C + s) * 64 + lane], v[s], acc[mb], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < C; g++) out[g] = acc[g / 4][g % 4];
}

// the same product for the four nodes of a round, one K step at a time: only four A operands are in flight at once
// (left to itself the scheduler hoists all 4 x C loads of a round above the first MFMA: 8 C registers per lane)
template <int RP>
__device__ __forceinline__ void mfma_prod4(const double *__restrict__ aop, const int (&jn)[4], const double (&v)[RP / 4],
                                           double (&out)[4][RP / 4], int lane)
{
    constexpr int C = RP / 4, MB = quad_mb(RP);
    v4d acc[4][MB];
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
#pragma unroll
        for (int mb = 0; mb < MB; mb++) acc[jj][mb] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < C; s++) {
        double av[4][MB];
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
            for (int mb = 0; mb < MB; mb++) av[jj][mb] = aop[(size_t)jn[jj] * quad_aop_node(RP) + (mb * C + s) * 64 + lane];
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
#pragma unroll
            for (int mb = 0; mb < MB; mb++) acc[jj][mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[jj][mb], v[s], acc[jj][mb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
#pragma unroll
        for (int g = 0; g < C; g++) out[jj][g] = acc[jj][g / 4][g % 4];
}

template <int C>
__device__ __forceinline__ double dot_c(const double (&a)[C], const double (&b)[C])
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < C; i++) s = fma(a[i], b[i], s);
    return s;
}

// NWV wavefronts per workgroup share the staged cores; registers per lane <= 512 / (NWV / 4)
template <class Model, int RP, int K, int NWV, bool ONEPASS>
__global__ void __launch_bounds__(64 * NWV, 1)
    k_fiber_quad(const KArgs A, const double *__restrict__ ro, const int32_t *__restrict__ idx, double *__restrict__ outv,
                 int32_t *__restrict__ uidx, int32_t *__restrict__ absorbed)
{
    constexpr int D = Model::D, S = 2 * D + 1, C = RP / 4;
    constexpr int NL = 2 * K, NR = 2 * (D - 1 - K);
    static_assert(RP % 4 == 0 && RP >= 4 && RP <= 32, "padded rank must be a multiple of 4");
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, q = lane >> 4, t = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = A.N;
    const int bck = A.bctype[K];
    double *sM = smem;                                // staged core of the current level
    // this wave's node values v_j: [N][16] with two passes; with one pass a ring of 16 nodes (a round is finalised one round
    // after it was computed and looks one node back and ahead) plus two slots for the periodic wrap (v_1, v_{N-2})
    constexpr int SVROWS = 18;
    double *sV = smem + A.quad_sv_off + wv * ((ONEPASS ? SVROWS : N) * 16);
    auto svslot = [&](int j) __attribute__((always_inline)) -> int {
        if constexpr (!ONEPASS) return j;
        else {
            const bool wrap = (bck == C3SC_PERIODIC);
            return (wrap && j == 1) ? 16 : ((wrap && j == N - 2) ? 17 : (j & 15));
        }
    };
    CandLds<Model> cr;
    {
        CandRegs<Model> cr0;
        cr0.load(A, ro);
        cr.fill(smem + A.tbl_off, cr0, A.ncand); // every wave writes the same rows
        __syncthreads();
    }
    unsigned st = 0;
    const long per_tile = 16L * NWV, ntiles = (A.F + per_tile - 1) / per_tile;
    const double *aopK = ro + A.quad_aop_off[K];

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long f_raw = tile * per_tile + wv * 16 + t;
        const bool flive = f_raw < A.F;
        const long f = flive ? f_raw : A.F - 1;
        // ---- fiber description (process_fibers_neighbor, fixed dims): identical in the four lanes of a fiber
        int fi[D];
        double x[D];
        bool fiber_abs = false;
#pragma unroll
        for (int m = 0; m < D; m++) {
            fi[m] = (m == K) ? 0 : idx[f * D + m];
            int lo, hi;
            const bool face = fixed_neighbors(fi[m], A.ngrid[m], A.bctype[m], lo, hi);
            if (m != K) fiber_abs = fiber_abs || face;
            x[m] = ro[A.xg_off[m] + fi[m]];
        }
        double tvf[Model::NTAB > 0 ? Model::NTAB : 1]; // tables not indexed by dim K are constants of the fiber
        table_values<Model>(A, ro, fi, tvf);

        double XL[1 + NL][C], XR[1 + NR][C]; // [0] = L / R, [1 + 2i + s] = (-,+) neighbour vector of dim K-1-i / K+1+i
#pragma unroll
        for (int s = 0; s < 1 + NL; s++)
#pragma unroll
            for (int i = 0; i < C; i++) XL[s][i] = 0.0;
#pragma unroll
        for (int s = 0; s < 1 + NR; s++)
#pragma unroll
            for (int i = 0; i < C; i++) XR[s][i] = 0.0;

        // ---- fold the suffix side: levels m = D-1 .. K+1
        if constexpr (K < D - 1) {
            int nlive = 1; // vectors in XR that exist so far (R counts)
#pragma nounroll
            for (int m = D - 1; m > K; m--) {
                const bool edge = (m == D - 1);
                const int elems = edge ? RP : RP * RP, ns = quad_stride(elems);
                __syncthreads();
                if (!(A.dbg & 512)) stage_padded(sM, ro + (edge ? A.core_off[m] : A.quad_coreT_off[m]), A.ngrid[m], elems);
                __syncthreads();
                const int nd = idx[f * D + m];
                int lo, hi;
                (void)fixed_neighbors(nd, A.ngrid[m], A.bctype[m], lo, hi);
                double T0[C], T1[C];
#pragma unroll
                for (int i = 0; i < C; i++) { T0[i] = 0.0; T1[i] = 0.0; }
                if (edge) { // r x 1 column core
#pragma unroll
                    for (int i = 0; i < C; i++) {
                        XR[0][i] = sM[nd * ns + q * C + i];
                        T0[i] = sM[lo * ns + q * C + i];
                        T1[i] = sM[hi * ns + q * C + i];
                    }
                } else if (!(A.dbg & 256)) {
                    apply_quad_pair<RP>(sM + lo * ns, sM + hi * ns, XR[0], T0, T1, q);
                    apply_live<RP, 1 + NR, 0>(sM + nd * ns, XR, nlive, q);
                }
#pragma unroll
                for (int s = NR; s >= 3; s--)
#pragma unroll
                    for (int i = 0; i < C; i++) XR[s][i] = XR[s - 2][i];
#pragma unroll
                for (int i = 0; i < C; i++) { XR[1][i] = T0[i]; XR[2][i] = T1[i]; }
                nlive += 2;
            }
        }
        // ---- fold the prefix side: levels m = 0 .. K-1 (the staged matrix is the transposed core = the arena layout)
        if constexpr (K > 0) {
            int nlive = 1;
#pragma nounroll
            for (int m = 0; m < K; m++) {
                const bool edge = (m == 0);
                const int elems = edge ? RP : RP * RP, ns = quad_stride(elems);
                __syncthreads();
                if (!(A.dbg & 512)) stage_padded(sM, ro + A.core_off[m], A.ngrid[m], elems);
                __syncthreads();
                const int nd = idx[f * D + m];
                int lo, hi;
                (void)fixed_neighbors(nd, A.ngrid[m], A.bctype[m], lo, hi);
                double T0[C], T1[C];
#pragma unroll
                for (int i = 0; i < C; i++) { T0[i] = 0.0; T1[i] = 0.0; }
                if (edge) { // 1 x r row core
#pragma unroll
                    for (int i = 0; i < C; i++) {
                        XL[0][i] = sM[nd * ns + q * C + i];
                        T0[i] = sM[lo * ns + q * C + i];
                        T1[i] = sM[hi * ns + q * C + i];
                    }
                } else if (!(A.dbg & 256)) {
                    apply_quad_pair<RP>(sM + lo * ns, sM + hi * ns, XL[0], T0, T1, q);
                    apply_live<RP, 1 + NL, 0>(sM + nd * ns, XL, nlive, q);
                }
#pragma unroll
                for (int s = NL; s >= 3; s--)
#pragma unroll
                    for (int i = 0; i < C; i++) XL[s][i] = XL[s - 2][i];
#pragma unroll
                for (int i = 0; i < C; i++) { XL[1][i] = T0[i]; XL[2][i] = T1[i]; }
                nlive += 2;
            }
        }

        // ---- c / a of node jn for this lane's fiber (own components)
        auto node_c = [&](int jn, double (&c)[C]) __attribute__((always_inline)) {
            if constexpr (K == D - 1) {
#pragma unroll
                for (int i = 0; i < C; i++) c[i] = ro[A.core_off[K] + (size_t)jn * RP + q * C + i];
            } else if constexpr (K > 0) {
                mfma_prod<RP>(aopK + (size_t)jn * quad_aop_node(RP), XR[0], c, lane);
            }
        };
        auto node_a = [&](int jn, double (&a)[C]) __attribute__((always_inline)) {
            if constexpr (K == 0) {
#pragma unroll
                for (int i = 0; i < C; i++) a[i] = ro[A.core_off[K] + (size_t)jn * RP + q * C + i];
            } else if constexpr (K < D - 1) {
                mfma_prod<RP>(aopK + (size_t)jn * quad_aop_node(RP) + quad_aop_node(RP) / 2, XL[0], a, lane);
            }
        };

        const bool forced = A.forced != nullptr; // wave-uniform
        // stencil of a round of four nodes, every dimension but K: lane (q, t) ends with the values of node j0 + q of fiber t;
        // with SELF the node values v_j = L G_k[j] R of the round are formed as well (from the c / a already in registers)
        // and put into this wave's LDS row
        auto stencil = [&](int j0, double (&V)[S], auto self_tag) __attribute__((always_inline)) {
            constexpr bool SELF = decltype(self_tag)::value;
            double vs[4] = {0.0, 0.0, 0.0, 0.0};
            const int jn[4] = {min(j0, N - 1), min(j0 + 1, N - 1), min(j0 + 2, N - 1), min(j0 + 3, N - 1)};
            if constexpr (K > 0) {
                double c[4][C];
                if constexpr (K < D - 1) mfma_prod4<RP>(aopK, jn, XR[0], c, lane);
                else {
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) node_c(jn[jj], c[jj]);
                }
#pragma unroll
                for (int i = 0; i < K; i++) {
                    const int m = K - 1 - i;
#pragma unroll
                    for (int s = 0; s < 2; s++)
                        V[2 * m + s] = reduce4(dot_c<C>(XL[1 + 2 * i + s], c[0]), dot_c<C>(XL[1 + 2 * i + s], c[1]),
                                               dot_c<C>(XL[1 + 2 * i + s], c[2]), dot_c<C>(XL[1 + 2 * i + s], c[3]));
                }
                if constexpr (SELF) {
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) vs[jj] = dot_c<C>(XL[0], c[jj]);
                }
            }
            __builtin_amdgcn_sched_barrier(0); // the c products and their dots retire before the a products start
            if constexpr (K < D - 1) {
                double a[4][C];
                if constexpr (K > 0) mfma_prod4<RP>(aopK + quad_aop_node(RP) / 2, jn, XL[0], a, lane);
                else {
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) node_a(jn[jj], a[jj]);
                }
#pragma unroll
                for (int i = 0; i < D - 1 - K; i++) {
                    const int m = K + 1 + i;
#pragma unroll
                    for (int s = 0; s < 2; s++)
                        V[2 * m + s] = reduce4(dot_c<C>(a[0], XR[1 + 2 * i + s]), dot_c<C>(a[1], XR[1 + 2 * i + s]),
                                               dot_c<C>(a[2], XR[1 + 2 * i + s]), dot_c<C>(a[3], XR[1 + 2 * i + s]));
                }
                if constexpr (SELF && K == 0) {
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) vs[jj] = dot_c<C>(a[jj], XR[0]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SELF) {
                const double v = reduce4(vs[0], vs[1], vs[2], vs[3]);
                if (j0 + q < N) sV[svslot(j0 + q) * 16 + t] = v;
            }
        };
        // boundary flags and control minimisation of node j0 + q of fiber t from its stencil (the dim-K entries come from sV)
        auto finalize = [&](int j0, double (&V)[S]) __attribute__((always_inline)) {
            const bool nlive = (j0 + q < N);
            const int j = nlive ? j0 + q : N - 1;
            // node coordinates, obstacle / face / end-point flags (nodeutil.c:496-624)
            x[K] = ro[A.xg_off[K] + j];
            int ab = in_obstacle<D>(A, ro, x) ? -1 : 0;
            if (fiber_abs) ab = 1;
            int lo, hi;
            ab = vary_neighbors(j, N, bck, ab, lo, hi, A.cends);
            V[2 * K] = sV[svslot(lo) * 16 + t];
            V[2 * K + 1] = sV[svslot(hi) * 16 + t];
            V[2 * D] = sV[svslot(j) * 16 + t];
            double tv[Model::NTAB > 0 ? Model::NTAB : 1];
            tv[0] = 0.0;
#pragma unroll
            for (int tt = 0; tt < Model::NTAB; tt++) tv[tt] = (Model::tab_dim(tt) == K) ? ro[A.tab_off[tt] + j] : tvf[tt];
            int ui = 0;
            const int fu = forced ? A.forced[(size_t)f * N + j] : -1;
            double val = V[0] + V[2 * D - 1];
            if (!(A.dbg & 1024)) val = node_backup<Model, 1, 1, CandLds<Model>, false>(A, ro, x, tv, cr, V, ab, ui, st, forced, fu);
            if (nlive && flive) {
                outv[(size_t)f * N + j] = val;
                if (uidx) uidx[(size_t)f * N + j] = ui;
                if (absorbed) absorbed[(size_t)f * N + j] = ab;
            }
        };
        wave_sync(); // the previous tile's readers of sV are done
        if constexpr (ONEPASS) {
            // one pass: a round's stencils wait in registers until the next round has produced the value of the node after
            // its last one (S more doubles per lane, no second set of products)
            if (bck == C3SC_PERIODIC && N > 5) { // node 0 needs v[N-2] long before the last round produces it
                double Vw[S];
                stencil((N - 2) & ~3, Vw, std::true_type{});
            }
            double Vp[S];
#pragma unroll
            for (int s = 0; s < S; s++) Vp[s] = 0.0;
            const int rounds = (N + 3) / 4;
            for (int r = 0; r <= rounds; r++) {
                double Vc[S];
#pragma unroll
                for (int s = 0; s < S; s++) Vc[s] = 0.0;
                if (r < rounds) stencil(4 * r, Vc, std::true_type{});
                wave_sync(); // this round's node values are visible to the wave
                if (r > 0) finalize(4 * (r - 1), Vp);
#pragma unroll
                for (int s = 0; s < S; s++) Vp[s] = Vc[s];
            }
        } else {
            // two passes: node values first (one product per node), then stencils and minimisation round by round
            for (int j0 = 0; j0 < ((A.dbg & 2048) ? 0 : N); j0 += 4) {
                double P[4];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int jn = min(j0 + jj, N - 1);
                    if constexpr (K == 0) {
                        double a[C];
                        node_a(jn, a);
                        P[jj] = dot_c<C>(a, XR[0]);
                    } else {
                        double c[C];
                        node_c(jn, c);
                        P[jj] = dot_c<C>(XL[0], c);
                    }
                }
                const double v = reduce4(P[0], P[1], P[2], P[3]);
                if (j0 + q < N) sV[svslot(j0 + q) * 16 + t] = v;
            }
            wave_sync();
            for (int j0 = 0; j0 < ((A.dbg & 4096) ? 0 : N); j0 += 4) {
                double V[S];
                stencil(j0, V, std::false_type{});
                finalize(j0, V);
            }
        }
    }
    if (st) atomicOr(A.status, st);
}

// ------------------------------------------------------------------------------------------------------------------
// k_fiber_quad_duo: the same lane layout with TWO wavefronts per 16 fibers.  At d = 10, rank 16 one wavefront holds
// 2(d-1)+2 = 20 folded vectors of four doubles per lane: with the MFMA operands and a round's stencil that is more than 256
// registers, the kernel above runs one wavefront per SIMD with the whole 512-entry file (and still spills), and FP64 VALU
// needs two wavefronts per SIMD to get past 40 % of its rate.  Here the neighbour vectors are divided between the two
// wavefronts of a pair (alternate dimensions by distance from K, the two sides offset so that the halves differ by one
// dimension at most); both keep the running prefix L and suffix R, so both form c and a of a node on the matrix cores
// (that pipe is idle otherwise) and each takes the dots of its own vectors.  Folding work per wavefront halves with the
// vectors.  Node loop in two passes: (1) node values v_j = L G_k[j] R, alternate rounds of four nodes per wavefront, into
// the pair's LDS row; (2) blocks of two rounds: both wavefronts form their half of both stencils, hand the half of the
// round they do not finalise to the partner through LDS, and finalise one round each at the same time.
__device__ inline void quad_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } // LDS traffic only: output stores stay in flight

// whole-workgroup copy of a core's LDS image (HBM layout = LDS layout) by LDS-DMA: 16 bytes per lane straight into LDS, no
// registers, nothing to wait for until the data is needed (the caller's s_waitcnt vmcnt(0) + barrier)
template <int NWV>
__device__ inline void quad_glds(double *dst, const double *img, int n_doubles)
{
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int pieces = n_doubles >> 1;
    const int lane = threadIdx.x & 63;
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 64;
    for (int base = wbase; base < pieces; base += 64 * NWV) {
        const int p = base + lane;
        if (p < pieces) __builtin_amdgcn_global_load_lds((glb_void *)(img + 2 * p), (lds_void *)((lds_char *)dst + 16 * base), 16, 0, 0);
    }
}

#ifndef FQD_PREFETCH
#define FQD_PREFETCH 1 // the next tile's indices and first core are requested during this tile's node loop (0: at the top of the tile)
#endif
#ifndef FQD_CG
#define FQD_CG 1 // control candidates in flight in the duo kernel's scan (3: 0.597 vs 0.588 ms on quad10d)
#endif
#ifndef FQD_NVB
#define FQD_NVB 3 // vectors per pass in the duo kernel's fold: a wavefront's live count is 1, 3, 5, ... (3: 0.656, 4: 0.661, 2: 0.668, 5: 0.690 ms on quad10d)
#endif
__host__ __device__ constexpr int duo_count(int n, int parity) { return n > parity ? (n - parity + 1) / 2 : 0; } // distances 0..n-1 of that parity

template <class Model, int RP, int K, int NWV, bool DBUF, int H>
__device__ __attribute__((always_inline)) inline void quad_duo_body(const KArgs &A, const double *__restrict__ ro,
                                                                   const int32_t *__restrict__ idx, double *__restrict__ outv,
                                                                   int32_t *__restrict__ uidx, int32_t *__restrict__ absorbed,
                                                                   double *smem, unsigned &st)
{
    constexpr int D = Model::D, S = 2 * D + 1, C = RP / 4;
    // own dimensions: left distance i = K-1-m with i % 2 == H, right distance i = m-K-1 with i % 2 == 1-H
    constexpr int NL = 2 * duo_count(K, H), NR = 2 * duo_count(D - 1 - K, 1 - H);
    const int lane = threadIdx.x & 63, q = lane >> 4, t = lane & 15;
    const int pw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7); // pair index in the workgroup
    const int N = A.N;
    const int bck = A.bctype[K];
    // two staging buffers, level l in buffer l & 1.  The address is formed by arithmetic on the LDS base: a pointer picked from an
    // array at run time loses its address space and every matrix read becomes a flat load (16 M vector-memory reads per launch
    // in the first version of this kernel, and the copy/compute overlap bought nothing)
    auto sMbuf = [&](int l) __attribute__((always_inline)) -> double * { return DBUF ? smem + (l & 1) * A.quad_m1_off : smem; };
    double *sV = smem + A.quad_sv_off + pw * (N * 16);      // node values of the pair's 16 fibers: [N][16]
    double *sX = smem + A.quad_x_off + pw * (2 * D * 64);   // half stencils on their way to the partner: [2D][64]
    // the pair's fiber indices [16][D], two sets: the next tile's arrive by LDS-DMA while this tile's nodes are finalised
    auto sIxb = [&](int par) __attribute__((always_inline)) -> int * {
        return reinterpret_cast<int *>(smem + A.quad_ix_off) + pw * (2 * 16 * D) + par * (16 * D);
    };
    constexpr int NLEV = D - 1, NRL = D - 1 - K;            // folding levels: l < NRL suffix side (m = D-1-l), then prefix (m = l-NRL)
    auto level_issue = [&](int l) __attribute__((always_inline)) {
        const bool right = l < NRL;
        const int m = right ? D - 1 - l : l - NRL;
        const int elems = (m == 0 || m == D - 1) ? RP : RP * RP;
        quad_glds<NWV>(sMbuf(l), A.img_base + (right ? A.quad_imgR_off[m] : A.quad_imgL_off[m]), A.ngrid[m] * quad_stride(elems));
    };
    CandLds<Model> cr;
    {
        CandRegs<Model> cr0;
        cr0.load(A, ro);
        cr.fill(smem + A.tbl_off, cr0, A.ncand); // every wave writes the same rows
        quad_barrier();
    }
    const long per_tile = 16L * (NWV / 2), ntiles = (A.F + per_tile - 1) / per_tile;
    const double *aopK = ro + A.quad_aop_off[K];
    // The 16 x D indices of a pair's fibers are contiguous in the batch: copied as they are (fibers past the end of the batch
    // repeat the last one), 4 bytes per lane, both wavefronts of the pair take part.
    auto idx_issue = [&](long tl, int par) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void glb_void;
        const long f0 = tl * per_tile + pw * 16;
        int *dst = sIxb(par);
#pragma unroll
        for (int p0 = H * 64; p0 < 16 * D; p0 += 128) {
            const int p = p0 + lane;
            if (p < 16 * D) {
                long fb = f0 + p / D;
                fb = fb < A.F ? fb : A.F - 1;
                __builtin_amdgcn_global_load_lds((glb_void *)(idx + fb * D + p % D), (lds_void *)(dst + p0), 4, 0, 0);
            }
        }
    };
    int par = 0;
    if constexpr (DBUF) { // the first tile's indices and first core; later ones are requested during the previous tile's node loop
        if (FQD_PREFETCH && (long)blockIdx.x < ntiles) { idx_issue(blockIdx.x, 0); level_issue(0); }
    }

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
        const long f_raw = tile * per_tile + pw * 16 + t;
        const bool flive = f_raw < A.F;
        const long f = flive ? f_raw : A.F - 1;
        // The fiber's indices live in LDS (the run-time level loops and the node loop read them there); coordinates, table values
        // and boundary flags are formed where a node is finalised instead of living in 30 registers through the fold.
        const int *sI = sIxb(par);
        if constexpr (DBUF && !FQD_PREFETCH) { idx_issue(tile, par); level_issue(0); }
        if constexpr (!DBUF) {
            if (H == 0 && q == 0) {
#pragma unroll
                for (int m = 0; m < D; m++) sIxb(par)[t * D + m] = idx[f * D + m];
            }
        }

        double XL[1 + NL][C], XR[1 + NR][C]; // [0] = L / R, then this wave's neighbour vectors, nearest own dimension first
#pragma unroll
        for (int s = 0; s < 1 + NL; s++)
#pragma unroll
            for (int i = 0; i < C; i++) XL[s][i] = 0.0;
#pragma unroll
        for (int s = 0; s < 1 + NR; s++)
#pragma unroll
            for (int i = 0; i < C; i++) XR[s][i] = 0.0;

        // ---- folding, level by level: the suffix side (m = D-1 .. K+1), then the prefix side (m = 0 .. K-1).  Level l's core
        // image goes into staging buffer l & 1 by LDS-DMA while level l-1 is being applied: one barrier per level, and the copy
        // (51 KB from L2 at rank 16) is hidden behind the products instead of stopping all eight wavefronts.
        if constexpr (K < D - 1) {
            int nlive = 1;
#pragma nounroll
            for (int l = 0; l < NRL; l++) {
                const int m = D - 1 - l;
                const bool mine = (((m - K - 1) & 1) == 1 - H) && NR > 0; // wave-uniform
                const bool edge = (m == D - 1);
                const int ns = quad_stride(edge ? RP : RP * RP);
                if constexpr (DBUF) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    quad_barrier(); // level l has landed for every wavefront, and every wavefront is done with the other buffer
                    if (l + 1 < NLEV && !(A.dbg & 512)) level_issue(l + 1);
                } else { // one buffer (two workgroups per CU cover each other's copies instead)
                    quad_barrier();
                    if (!(A.dbg & 512)) level_issue(l);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    quad_barrier();
                }
                const double *sM = sMbuf(l);
                const int nd = sI[t * D + m];
                int lo, hi;
                (void)fixed_neighbors(nd, A.ngrid[m], A.bctype[m], lo, hi);
                double T0[C], T1[C];
#pragma unroll
                for (int i = 0; i < C; i++) { T0[i] = 0.0; T1[i] = 0.0; }
                if (edge) {
#pragma unroll
                    for (int i = 0; i < C; i++) {
                        XR[0][i] = sM[nd * ns + q * C + i];
                        T0[i] = sM[lo * ns + q * C + i];
                        T1[i] = sM[hi * ns + q * C + i];
                    }
                } else if (!(A.dbg & 256)) {
                    if (mine) apply_quad_pair<RP>(sM + lo * ns, sM + hi * ns, XR[0], T0, T1, q);
                    apply_live<RP, 1 + NR, 0, FQD_NVB>(sM + nd * ns, XR, nlive, q);
                }
                if constexpr (NR > 0) {
                    if (mine) {
#pragma unroll
                        for (int s = NR; s >= 3; s--)
#pragma unroll
                            for (int i = 0; i < C; i++) XR[s][i] = XR[s - 2][i];
#pragma unroll
                        for (int i = 0; i < C; i++) { XR[1][i] = T0[i]; XR[2][i] = T1[i]; }
                        nlive += 2;
                    }
                }
            }
        }
        if constexpr (K > 0) {
            int nlive = 1;
#pragma nounroll
            for (int l = NRL; l < NLEV; l++) {
                const int m = l - NRL;
                const bool mine = (((K - 1 - m) & 1) == H) && NL > 0;
                const bool edge = (m == 0);
                const int ns = quad_stride(edge ? RP : RP * RP);
                if constexpr (DBUF) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    quad_barrier();
                    if (l + 1 < NLEV && !(A.dbg & 512)) level_issue(l + 1);
                } else { // one buffer (two workgroups per CU cover each other's copies instead)
                    quad_barrier();
                    if (!(A.dbg & 512)) level_issue(l);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    quad_barrier();
                }
                const double *sM = sMbuf(l);
                const int nd = sI[t * D + m];
                int lo, hi;
                (void)fixed_neighbors(nd, A.ngrid[m], A.bctype[m], lo, hi);
                double T0[C], T1[C];
#pragma unroll
                for (int i = 0; i < C; i++) { T0[i] = 0.0; T1[i] = 0.0; }
                if (edge) {
#pragma unroll
                    for (int i = 0; i < C; i++) {
                        XL[0][i] = sM[nd * ns + q * C + i];
                        T0[i] = sM[lo * ns + q * C + i];
                        T1[i] = sM[hi * ns + q * C + i];
                    }
                } else if (!(A.dbg & 256)) {
                    if (mine) apply_quad_pair<RP>(sM + lo * ns, sM + hi * ns, XL[0], T0, T1, q);
                    apply_live<RP, 1 + NL, 0, FQD_NVB>(sM + nd * ns, XL, nlive, q);
                }
                if constexpr (NL > 0) {
                    if (mine) {
#pragma unroll
                        for (int s = NL; s >= 3; s--)
#pragma unroll
                            for (int i = 0; i < C; i++) XL[s][i] = XL[s - 2][i];
#pragma unroll
                        for (int i = 0; i < C; i++) { XL[1][i] = T0[i]; XL[2][i] = T1[i]; }
                        nlive += 2;
                    }
                }
            }
        }

        // c / a of the four nodes of a round (own components): matrix cores for a middle core, the core itself at the ends
        auto round_c = [&](const int (&jn)[4], double (&c)[4][C]) __attribute__((always_inline)) {
            if constexpr (K == D - 1) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
#pragma unroll
                    for (int i = 0; i < C; i++) c[jj][i] = ro[A.core_off[K] + (size_t)jn[jj] * RP + q * C + i];
            } else mfma_prod4<RP>(aopK, jn, XR[0], c, lane);
        };
        auto round_a = [&](const int (&jn)[4], double (&a)[4][C]) __attribute__((always_inline)) {
            if constexpr (K == 0) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
#pragma unroll
                    for (int i = 0; i < C; i++) a[jj][i] = ro[A.core_off[K] + (size_t)jn[jj] * RP + q * C + i];
            } else mfma_prod4<RP>(aopK + quad_aop_node(RP) / 2, jn, XL[0], a, lane);
        };
        const bool forced = A.forced != nullptr;
        const int rounds = (N + 3) / 4;

        // Node values v_j = L G_k[j] R (the dim-K neighbours of a node and its own value) come out of the c products wavefront 0
        // forms anyway: it writes the values of a block's eight nodes into the pair's LDS row, wavefront 1 adds the one node after
        // the block (and, under a periodic boundary, node N-2 before the first block) with a single-node product.
        static_assert(H != 0 || (K > 0 ? NL > 0 : NR > 0), "wavefront 0 forms the product the node values come from");
        auto one_value = [&](int jx) __attribute__((always_inline)) { // lanes q == 0 write v[jx]
            double vs = 0.0;
            if constexpr (K == 0) {
#pragma unroll
                for (int i = 0; i < C; i++) vs = fma(ro[A.core_off[K] + (size_t)jx * RP + q * C + i], XR[0][i], vs);
            } else if constexpr (K == D - 1) {
#pragma unroll
                for (int i = 0; i < C; i++) vs = fma(XL[0][i], ro[A.core_off[K] + (size_t)jx * RP + q * C + i], vs);
            } else {
                double c1[C];
                mfma_prod<RP>(aopK + (size_t)jx * quad_aop_node(RP), XR[0], c1, lane);
                vs = dot_c<C>(XL[0], c1);
            }
            const double v = reduce4(vs, 0.0, 0.0, 0.0);
            if (q == 0) sV[jx * 16 + t] = v;
        };
        if constexpr (H == 1) {
            if (bck == C3SC_PERIODIC && N > 2) one_value(N - 2); // wave-uniform
        }

        // own half of the stencil of a round: lane (q, t) ends with the values of node j0 + q of fiber t
        auto stencil_own = [&](int j0, double (&V)[S]) __attribute__((always_inline)) {
            const int jn[4] = {min(j0, N - 1), min(j0 + 1, N - 1), min(j0 + 2, N - 1), min(j0 + 3, N - 1)};
            if constexpr (K > 0 && NL > 0) {
                double c[4][C];
                round_c(jn, c);
                if constexpr (H == 0) {
                    const double v = reduce4(dot_c<C>(XL[0], c[0]), dot_c<C>(XL[0], c[1]), dot_c<C>(XL[0], c[2]), dot_c<C>(XL[0], c[3]));
                    if (j0 + q < N) sV[(j0 + q) * 16 + t] = v;
                }
#pragma unroll
                for (int i = 0; i < K; i++) {
                    if ((i & 1) != H) continue;
                    const int m = K - 1 - i, slot = 1 + 2 * (i / 2);
#pragma unroll
                    for (int s = 0; s < 2; s++)
                        V[2 * m + s] = reduce4(dot_c<C>(XL[slot + s], c[0]), dot_c<C>(XL[slot + s], c[1]), dot_c<C>(XL[slot + s], c[2]),
                                               dot_c<C>(XL[slot + s], c[3]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (K < D - 1 && NR > 0) {
                double a[4][C];
                round_a(jn, a);
                if constexpr (H == 0 && K == 0) {
                    const double v = reduce4(dot_c<C>(a[0], XR[0]), dot_c<C>(a[1], XR[0]), dot_c<C>(a[2], XR[0]), dot_c<C>(a[3], XR[0]));
                    if (j0 + q < N) sV[(j0 + q) * 16 + t] = v;
                }
#pragma unroll
                for (int i = 0; i < D - 1 - K; i++) {
                    if ((i & 1) != 1 - H) continue;
                    const int m = K + 1 + i, slot = 1 + 2 * (i / 2);
#pragma unroll
                    for (int s = 0; s < 2; s++)
                        V[2 * m + s] = reduce4(dot_c<C>(a[0], XR[slot + s]), dot_c<C>(a[1], XR[slot + s]), dot_c<C>(a[2], XR[slot + s]),
                                               dot_c<C>(a[3], XR[slot + s]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto finalize = [&](int j0, double (&V)[S]) __attribute__((always_inline)) {
            const bool nlive = (j0 + q < N);
            const int j = nlive ? j0 + q : N - 1;
            int fi[D];
            double x[D];
            bool fiber_abs = false; // an absorbing face in a fixed dimension absorbs the whole fiber (nodeutil.c:515-523)
#pragma unroll
            for (int m = 0; m < D; m++) {
                fi[m] = (m == K) ? j : sI[t * D + m];
                x[m] = ro[A.xg_off[m] + fi[m]];
                if (m != K) {
                    int flo, fhi;
                    fiber_abs = fixed_neighbors(fi[m], A.ngrid[m], A.bctype[m], flo, fhi) || fiber_abs;
                }
            }
            double tvf[Model::NTAB > 0 ? Model::NTAB : 1];
            table_values<Model>(A, ro, fi, tvf);
            int ab = in_obstacle<D>(A, ro, x) ? -1 : 0;
            if (fiber_abs) ab = 1;
            int lo, hi;
            ab = vary_neighbors(j, N, bck, ab, lo, hi, A.cends);
            V[2 * K] = sV[lo * 16 + t];
            V[2 * K + 1] = sV[hi * 16 + t];
            V[2 * D] = sV[j * 16 + t];
            double tv[Model::NTAB > 0 ? Model::NTAB : 1];
            tv[0] = 0.0;
#pragma unroll
            for (int tt = 0; tt < Model::NTAB; tt++) tv[tt] = tvf[tt]; // fi[K] = j: tables indexed by dim K included
            int ui = 0;
            const int fu = forced ? A.forced[(size_t)f * N + j] : -1;
            double val = V[0] + V[2 * D - 1];
            if (!(A.dbg & 1024)) val = node_backup<Model, FQD_CG, FQD_CG, CandLds<Model>, false>(A, ro, x, tv, cr, V, ab, ui, st, forced, fu);
            if (nlive && flive) {
                outv[(size_t)f * N + j] = val;
                if (uidx) uidx[(size_t)f * N + j] = ui;
                if (absorbed) absorbed[(size_t)f * N + j] = ab;
            }
        };
        // is stencil entry e = 2m + s (m != K) formed by wavefront h?
        auto owner_of = [](int m) __attribute__((always_inline)) -> int { return m < K ? ((K - 1 - m) & 1) : 1 - ((m - K - 1) & 1); };

        // ---- node loop: blocks of two rounds
        for (int b = 0; 2 * b < ((A.dbg & 4096) ? 0 : rounds); b++) {
            double Vm[S];
#pragma unroll
            for (int s = 0; s < S; s++) Vm[s] = 0.0;
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int r = 2 * b + rr;
                if (r < rounds) { // wave-uniform
                    if (rr == H) stencil_own(4 * r, Vm);
                    else {
                        double Vh[S];
#pragma unroll
                        for (int s = 0; s < S; s++) Vh[s] = 0.0;
                        stencil_own(4 * r, Vh);
#pragma unroll
                        for (int m = 0; m < D; m++)
                            if (m != K && owner_of(m) == H) { sX[(2 * m) * 64 + lane] = Vh[2 * m]; sX[(2 * m + 1) * 64 + lane] = Vh[2 * m + 1]; }
                    }
                }
            }
            if constexpr (H == 1) {
                if (8 * b + 8 < N) one_value(8 * b + 8); // the right neighbour of the block's last node
            }
            quad_barrier();
            if constexpr (DBUF) { // every wavefront is past the fold: both staging buffers are free until the next tile
                if (FQD_PREFETCH && b == 0 && tile + gridDim.x < ntiles) { idx_issue(tile + gridDim.x, par ^ 1); level_issue(0); }
            }
            if (2 * b + H < rounds) {
#pragma unroll
                for (int m = 0; m < D; m++)
                    if (m != K && owner_of(m) != H) { Vm[2 * m] = sX[(2 * m) * 64 + lane]; Vm[2 * m + 1] = sX[(2 * m + 1) * 64 + lane]; }
                finalize(4 * (2 * b + H), Vm);
            }
            quad_barrier(); // the exchange rows (and, after the last block, the node values) may be overwritten
        }
    }
}

template <class Model, int RP, int K, int NWV, bool DBUF>
__global__ void __launch_bounds__(64 * NWV, DBUF ? 1 : 2)
    k_fiber_quad_duo(const KArgs A, const double *__restrict__ ro, const int32_t *__restrict__ idx, double *__restrict__ outv,
                     int32_t *__restrict__ uidx, int32_t *__restrict__ absorbed)
{
    static_assert(RP % 4 == 0 && RP >= 4 && RP <= 32 && NWV % 2 == 0, "padded rank must be a multiple of 4, wavefronts come in pairs");
    extern __shared__ double smem_duo[];
    unsigned st = 0;
    const int h = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 1);
    if (h == 0) quad_duo_body<Model, RP, K, NWV, DBUF, 0>(A, ro, idx, outv, uidx, absorbed, smem_duo, st);
    else quad_duo_body<Model, RP, K, NWV, DBUF, 1>(A, ro, idx, outv, uidx, absorbed, smem_duo, st);
    if (st) atomicOr(A.status, st);
}

} // namespace c3sc
