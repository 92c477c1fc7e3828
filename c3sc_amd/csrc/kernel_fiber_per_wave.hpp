// kernel_fiber_per_wave.hpp -- "one fiber per wavefront, one node per lane" Bellman kernel.
//
// Mapping (gfx950, wave64):
//   * a 256-thread workgroup = 4 wavefronts; every wavefront owns one fiber at a time and walks
//     the batch with a grid stride (persistent workgroups);
//   * lane l (+64q for q < NPL) is node j of the fiber, so everything that is fixed along the
//     fiber -- the fixed indices, the cores G_m[i_m] of the dims m != k -- is wave-uniform and is
//     fetched with scalar loads into SGPRs, and the r x r chain products run as v_fma_f64 with an
//     SGPR matrix operand (no LDS / VGPR traffic for the matrices);
//   * the varying core G_k (N_k matrices, one per lane) is staged once per workgroup in LDS with an
//     odd row stride so that the per-lane matrix reads are bank-conflict free;
//   * the per-fiber row/column vectors (prefix L, suffix R and the neighbour vectors
//     q_m^{+-} = L_{m-1} G_m[i_m +- 1],  z_m^{+-} = G_m[i_m +- 1] R_{m+1}) are formed once per fiber by
//     3*RP lanes and kept in a per-wave LDS scratch; the node loop reads them as LDS broadcasts.
//
// FT algebra (restating valuef_eval_fiber_ind_nn, src/valuefunc.c:369-585): with
//   c_j^(m) = G_m[i_m] ... G_{k-1}[i_{k-1}] G_k[j] R      (suffix side, pushed towards dim 0)
//   a_j^(m) = L G_k[j] G_{k+1}[i_{k+1}] ... G_m[i_m]      (prefix side, pushed towards dim d-1)
// the value at node j is L . c_j^(k), the (-,+) neighbour in a dim m < k is q_m^{-+} . c_j^(m+1) and
// in a dim m > k it is a_j^(m-1) . z_m^{-+}.  (The reference multiplies the neighbour core into every
// node vector, :535-544/:569-578; contracting it with the fiber-constant prefix first is the same
// number with r^2 -> r work per node.)
#pragma once
#include "kernel_common.hpp"

namespace c3sc {

template <int RP>
__device__ inline void gemv_n_sgpr(const double *__restrict__ G, double (&c)[RP])
{ // c <- G c, G wave-uniform (scalar loads), col-major a + b*RP   (valuefunc.c:504-506)
    double r[RP];
#pragma unroll
    for (int a = 0; a < RP; a++) r[a] = 0.0;
#pragma unroll
    for (int b = 0; b < RP; b++)
#pragma unroll
        for (int a = 0; a < RP; a++) r[a] = fma(G[a + b * RP], c[b], r[a]);
#pragma unroll
    for (int a = 0; a < RP; a++) c[a] = r[a];
}

template <int RP>
__device__ inline void gemv_t_sgpr(const double *__restrict__ G, double (&v)[RP])
{ // v <- v G   (valuefunc.c:491-493)
    double r[RP];
#pragma unroll
    for (int b = 0; b < RP; b++) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < RP; a++) s = fma(v[a], G[a + b * RP], s);
        r[b] = s;
    }
#pragma unroll
    for (int b = 0; b < RP; b++) v[b] = r[b];
}

template <int RP>
__device__ inline double dot_lds(const double *s, const double (&v)[RP])
{
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < RP; a++) acc = fma(s[a], v[a], acc);
    return acc;
}

// LDS stride (in doubles) of one node's matrix of the varying core: odd => conflict-free ds_read_b64
__host__ __device__ constexpr int kcore_stride(int rp, bool edge) { return (edge ? rp : rp * rp) | 1; }

// (256, 2): two workgroups per CU = 2 waves per SIMD, i.e. at most 256 registers per lane; without the bound the
// rank-16 instantiations take 256 + a few AGPRs and drop to one wave per SIMD (quad10d: 9.7e8 -> 7.0e8 nodes/s)
template <class Model, int RP, int NPL, bool STENCIL, bool BOX = false, bool STAGED = true>
__global__ void __launch_bounds__(256, (RP >= 20 ? 1 : 2)) // rank 20 is LDS-bound to one workgroup per CU anyway
    k_fiber_per_wave(const KArgs A, const double *__restrict__ ro, const int32_t *__restrict__ idx, double *__restrict__ outv,
                     int32_t *__restrict__ uidx, int32_t *__restrict__ absorbed, const int32_t *__restrict__ nbf,
                     const int32_t *__restrict__ nbv, const double *__restrict__ tbl, const double *__restrict__ tcost)
{
    constexpr int D = Model::D;
    constexpr int S = 2 * D + 1;
    constexpr int WS = 4 * RP + 2 * D * RP + 64 * NPL; // per-wave scratch (doubles)
    if (A.skip != nullptr && __builtin_amdgcn_readfirstlane(*A.skip) != 0) return; // the caller already holds these values (cross_device.hip)
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *sP = smem + wv * WS;     // [2][RP] prefix ping-pong
    double *sS = sP + 2 * RP;        // [2][RP] suffix ping-pong
    double *sNB = sS + 2 * RP;       // [D][2][RP] neighbour vectors q / z
    double *sV = sNB + 2 * D * RP;   // [64*NPL] node values of the fiber
    double *sK = smem + 4 * WS;      // staged varying core
    const int k = A.k, N = A.N;
    const bool kedge = (k == 0) || (k == D - 1);
    const int kstr = kcore_stride(RP, kedge);
    const int kelems = kedge ? RP : RP * RP;

    // ---- stage core k (shared by every fiber of this launch) into LDS, coalesced.  STAGED = false: the core is too
    // large for the CU's LDS (N x RP^2 doubles, e.g. rank 20 on a 100-node dimension); every lane then reads its own
    // node's matrix from global memory (L2-resident), the rest of the kernel is unchanged.
    if constexpr (STAGED) {
        const double *src = ro + A.core_off[k];
        const int total = N * kelems;
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const int j = e / kelems, w = e - j * kelems;
            sK[j * kstr + w] = src[e];
        }
    }
    __syncthreads();

    const int sel = lane / RP, bb = lane - sel * RP; // (vector, component) role in the per-fiber setup
    // candidate table in LDS (read with wave-uniform addresses): this kernel HAS lane-divergent control flow (lanes are
    // nodes), so a table kept in VGPR lanes could be corrupted by a spill reloaded under a partial EXEC mask
    CandLds<Model> cr;
    cr.tb = nullptr;
    if constexpr (!STENCIL && !Model::IS_TABLE) {
        for (int c0 = 0; c0 < A.ncand; c0 += 64) { // every wave writes the same rows; 64 candidates per pass
            CandRegs<Model> cr0;
            cr0.load(A, ro, c0);
            cr.fill(smem + A.tbl_off, cr0, A.ncand, c0);
        }
        __syncthreads();
    }
    unsigned st = 0;

    for (long f = (long)blockIdx.x * 4 + wv; f < A.F; f += (long)gridDim.x * 4) {
        // ---------------- wave-uniform fiber description (process_fibers_neighbor, fixed dims)
        int fi[D], nbm[D], nbp[D];
        bool fiber_abs = false;
#pragma unroll
        for (int m = 0; m < D; m++) {
            fi[m] = idx[f * D + m];
            if (m == k) fi[m] = 0;
            const bool face = fixed_neighbors(fi[m], A.ngrid[m], A.bctype[m], nbm[m], nbp[m]);
            if (m != k) fiber_abs = fiber_abs || face;
        }
        if constexpr (STENCIL) {
            // caller-supplied neighbour indices (the literal valuef_eval_fiber_ind_nn interface,
            // valuefunc.c:369-371: neighbors[2(d-1)] skips dim k)
            if (nbf) {
#pragma unroll
                for (int m = 0; m < D; m++) {
                    if (m != k) {
                        const int on = 2 * (m < k ? m : m - 1);
                        nbm[m] = nbf[f * 2 * (D - 1) + on];
                        nbp[m] = nbf[f * 2 * (D - 1) + on + 1];
                    }
                }
            }
        }

        // ---------------- prefix side: L = G_0[i_0] ... G_{k-1}[i_{k-1}],  q_m = L_{m-1} G_m[nb]
        int cur = 0;
        if (k > 0) {
            if (sel < 3) {
                const int node = sel == 0 ? fi[0] : (sel == 1 ? nbm[0] : nbp[0]);
                const double v = ro[A.core_off[0] + (size_t)node * RP + bb];
                if (sel == 0) sP[bb] = v;
                else sNB[(0 * 2 + sel - 1) * RP + bb] = v;
            }
            wave_sync();
#pragma unroll
            for (int m = 1; m < D - 1; m++) {
                if (m < k) {
                    if (sel < 3) {
                        const int node = sel == 0 ? fi[m] : (sel == 1 ? nbm[m] : nbp[m]);
                        const double *col = ro + A.core_off[m] + ((size_t)node * RP + bb) * RP; // column bb
                        double acc = 0.0;
#pragma unroll
                        for (int a = 0; a < RP; a++) acc = fma(sP[cur * RP + a], col[a], acc);
                        if (sel == 0) sP[(cur ^ 1) * RP + bb] = acc;
                        else sNB[(m * 2 + sel - 1) * RP + bb] = acc;
                    }
                    wave_sync();
                    cur ^= 1;
                }
            }
        }
        const double *Lv = sP + cur * RP;

        // ---------------- suffix side: R = G_{k+1}[i] ... G_{d-1}[i],  z_m = G_m[nb] R_{m+1}
        int curs = 0;
        if (k < D - 1) {
            if (sel < 3) {
                const int node = sel == 0 ? fi[D - 1] : (sel == 1 ? nbm[D - 1] : nbp[D - 1]);
                const double v = ro[A.core_off[D - 1] + (size_t)node * RP + bb];
                if (sel == 0) sS[bb] = v;
                else sNB[((D - 1) * 2 + sel - 1) * RP + bb] = v;
            }
            wave_sync();
#pragma unroll
            for (int m = D - 2; m >= 1; m--) {
                if (m > k) {
                    if (sel < 3) {
                        const int node = sel == 0 ? fi[m] : (sel == 1 ? nbm[m] : nbp[m]);
                        const double *row = ro + A.core_off[m] + (size_t)node * RP * RP + bb; // row bb, stride RP
                        double acc = 0.0;
#pragma unroll
                        for (int b = 0; b < RP; b++) acc = fma(row[b * RP], sS[curs * RP + b], acc);
                        if (sel == 0) sS[(curs ^ 1) * RP + bb] = acc;
                        else sNB[(m * 2 + sel - 1) * RP + bb] = acc;
                    }
                    wave_sync();
                    curs ^= 1;
                }
            }
        }
        const double *Rv = sS + curs * RP;

        // ---------------- pass 1: the varying core.  c = G_k[j] R, a = L G_k[j], v = L c
        double c[NPL][RP], aa[NPL][RP], vself[NPL];
#pragma unroll
        for (int q = 0; q < NPL; q++) {
            const int j = lane + 64 * q;
            const int jj = j < N ? j : N - 1; // clamp idle lanes onto a valid node
            const double *g;
            if constexpr (STAGED) g = sK + jj * kstr;
            else g = ro + A.core_off[k] + (size_t)jj * kelems;
            if (k == 0) { // first core: 1 x r row
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < RP; b++) {
                    aa[q][b] = g[b];
                    c[q][b] = 0.0;
                    v = fma(g[b], Rv[b], v);
                }
                vself[q] = v;
            } else if (k == D - 1) { // last core: r x 1 column
                double v = 0.0;
#pragma unroll
                for (int a = 0; a < RP; a++) {
                    c[q][a] = g[a];
                    aa[q][a] = 0.0;
                    v = fma(Lv[a], g[a], v);
                }
                vself[q] = v;
            } else {
#pragma unroll
                for (int a = 0; a < RP; a++) c[q][a] = 0.0;
#pragma unroll
                for (int b = 0; b < RP; b++) {
                    const double rb = Rv[b];
                    double s = 0.0;
#pragma unroll
                    for (int a = 0; a < RP; a++) {
                        const double gv = g[a + b * RP];
                        c[q][a] = fma(gv, rb, c[q][a]);
                        s = fma(Lv[a], gv, s);
                    }
                    aa[q][b] = s;
                }
                vself[q] = dot_lds<RP>(Lv, c[q]);
            }
            if (j < N) sV[j] = vself[q];
        }
        wave_sync();

        // ---------------- pass 2: neighbours, boundary flags, control minimisation
#pragma unroll
        for (int q = 0; q < NPL; q++) {
            const int j = lane + 64 * q;
            const bool live = j < N;
            const int jj = live ? j : N - 1;
            double V[S];
            V[2 * D] = vself[q];
            // dims before k: walk the suffix-side vector towards dim 0
#pragma unroll
            for (int m = D - 2; m >= 1; m--) {
                if (m < k) {
                    V[2 * m] = dot_lds<RP>(sNB + (m * 2 + 0) * RP, c[q]);
                    V[2 * m + 1] = dot_lds<RP>(sNB + (m * 2 + 1) * RP, c[q]);
                    gemv_n_sgpr<RP>(ro + A.core_off[m] + (size_t)fi[m] * RP * RP, c[q]);
                }
            }
            if (k > 0) {
                V[0] = dot_lds<RP>(sNB + 0 * RP, c[q]);
                V[1] = dot_lds<RP>(sNB + 1 * RP, c[q]);
            }
            // dims after k: walk the prefix-side vector towards dim d-1
#pragma unroll
            for (int m = 1; m < D - 1; m++) {
                if (m > k) {
                    V[2 * m] = dot_lds<RP>(sNB + (m * 2 + 0) * RP, aa[q]);
                    V[2 * m + 1] = dot_lds<RP>(sNB + (m * 2 + 1) * RP, aa[q]);
                    gemv_t_sgpr<RP>(ro + A.core_off[m] + (size_t)fi[m] * RP * RP, aa[q]);
                }
            }
            if (k < D - 1) {
                V[2 * (D - 1)] = dot_lds<RP>(sNB + ((D - 1) * 2 + 0) * RP, aa[q]);
                V[2 * (D - 1) + 1] = dot_lds<RP>(sNB + ((D - 1) * 2 + 1) * RP, aa[q]);
            }

            // node coordinates, obstacle / face / end-point flags (nodeutil.c:496-624)
            double x[D];
            int ix[D];
#pragma unroll
            for (int m = 0; m < D; m++) {
                ix[m] = (m == k) ? jj : fi[m];
                x[m] = ro[A.xg_off[m] + ix[m]];
            }
            int ab = in_obstacle<D>(A, ro, x) ? -1 : 0;
            if (fiber_abs) ab = 1;
            int lo, hi;
            ab = vary_neighbors(jj, N, A.bctype[k], ab, lo, hi, A.cends);
            if constexpr (STENCIL) {
                if (nbv) {
                    lo = nbv[(f * N + jj) * 2];
                    hi = nbv[(f * N + jj) * 2 + 1];
                }
            }
            const double vlo = sV[lo], vhi = sV[hi];
#pragma unroll
            for (int m = 0; m < D; m++)
                if (m == k) { V[2 * m] = vlo; V[2 * m + 1] = vhi; }

            if constexpr (STENCIL) {
                if (live) {
                    double *o = outv + ((size_t)f * N + j) * S;
#pragma unroll
                    for (int s = 0; s < S; s++) o[s] = V[s];
                    if (absorbed) absorbed[(size_t)f * N + j] = ab;
                }
            } else {
                int ui;
                double val;
                const bool forced = A.forced != nullptr; // wave-uniform
                const int fu = forced ? A.forced[(size_t)f * N + jj] : -1;
                if constexpr (Model::IS_TABLE) {
                    const size_t node = (size_t)f * N + jj;
                    val = node_backup_tables<D>(A, tbl + node * A.ncand * S, tcost + node * 2, V, ab, ui, st, forced, fu);
                } else if constexpr (BOX) { // continuous controls in a box: a separate instantiation (the minimiser
                                            // costs ~60 VGPRs, which halves the occupancy of the rank-16 kernels)
                    static_assert(Model::NCF == 0 || requires { Model::CF_FROM_U; }, "box minimiser: per-candidate features must be computable from u on the device");
                    double tv[Model::NTAB > 0 ? Model::NTAB : 1];
                    table_values<Model>(A, ro, ix, tv);
                    {
                        double uo[Model::DU];
                        const bool fcd = A.forced_u != nullptr;
                        val = node_backup_box<Model>(A, x, tv, V, ab, uo, st, fcd,
                                                     fcd ? A.forced_u + ((size_t)f * N + jj) * Model::DU : nullptr);
                        ui = -1;
                        if (live && A.uopt)
                            for (int i = 0; i < Model::DU; i++) A.uopt[((size_t)f * N + j) * Model::DU + i] = uo[i];
                    }
                } else {
                    double tv[Model::NTAB > 0 ? Model::NTAB : 1];
                    table_values<Model>(A, ro, ix, tv);
                    val = node_backup<Model, 1, 1, CandLds<Model>>(A, ro, x, tv, cr, V, ab, ui, st, forced, fu);
                }
                if (A.memo_keys) { // wave-uniform: node memo of the device-resident cross iterations
                    int ins = 0, ovf = 0;
                    if (live && (A.memo_mode == 0 || ab == 0)) {
                        unsigned long long id = 0;
#pragma unroll
                        for (int m = 0; m < D; m++) id += (unsigned long long)ix[m] * (unsigned long long)A.memo_stride[m];
                        const double mv = memo_merge(A.memo_keys, A.memo_vals, A.memo_capmask, A.memo_shift, A.memo_epoch_bits, id,
                                                     A.memo_mode == 0 ? val : (double)ui, ins, ovf);
                        if (A.memo_mode == 0) val = mv;
                        else ui = (int)mv;
                    }
                    const unsigned long long mk = __ballot(ins);
                    if (mk != 0 && lane == 0) atomicAdd(&A.memo_counters[0], (unsigned long long)__popcll(mk));
                    if (__any(ovf) && lane == 0) atomicExch(&A.memo_counters[3], 1ull);
                }
                if (live) {
                    outv[(size_t)f * N + j] = val;
                    if (uidx) uidx[(size_t)f * N + j] = ui;
                    if (absorbed) absorbed[(size_t)f * N + j] = ab;
                }
            }
        }
        wave_sync(); // sV / sNB / sP are rewritten by the next fiber
    }
    if (st) atomicOr(A.status, st);
}

} // namespace c3sc
