// cross_device.hip -- device-resident core steps of the TT-cross approximation that drives the Bellman sweep.
//
// The reference hands bellman_vi to C3's cross approximation (valuefunc.c:603-767); every core step of it asks for
// r_k r_{k+1} fibers, factors the N r x r matrix of their values and derives the next index set from the pivots -- a
// sequential chain of ~14 small steps per cross iteration ("granularity mismatch").  With the fibers on
// the GPU and the dense algebra + node memo on the host, a car7d sweep was host-bound (7.4 ms for 85 000 node backups).
// Here a whole cross iteration (left-to-right and right-to-left half sweep) is enqueued on one stream without a host round
// trip; per core step:
//
//   k_cross_idx    fiber index list  I_k x J_k                                   (cross_eval_core of c3sc_cross.c)
//   Bellman kernel c3sc_hip_bellman_fibers[_box] on the device buffers           (bellman.c:1295-1423 batched)
//   k_cross_memo   node memo: open addressing keyed by the mixed-radix node id and the sweep's epoch, first value stays
//                  (bellman.c:1333-1353, 1412-1417; hashgrid.c:252-279)
//   k_cross_core   one workgroup: tall LU with row pivoting of the fiber matrix, B = L inv(L[rows]), maxvol row swaps until
//                  max |B| <= 1.01, interpolatory core + next index set         (cross_sweep_lr / _rl of c3sc_cross.c)
//
// The numerics are those of the host driver (c3sc_amd/host/c3sc_cross.c: lu_maxvol) to the bit: the factorisation reorders no
// floating-point sum, contraction is off in k_cross_core, and pivot searches compare integer keys (magnitude with 22 mantissa
// bits dropped, ties to the lowest index) -- tests/test_solver_loops.py holds the two paths to identical cores.
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "ctx.hpp"

using namespace c3sc;

namespace {

constexpr int NT = 512;                       // threads of the core-step workgroup
constexpr int MAXROWS = 8192;                 // rows of a fiber matrix (N r): 16 per thread
constexpr unsigned long long IDX_BITS = 22;   // low bits of a pivot-search key hold the (inverted) index
constexpr unsigned long long IDX_MASK = (1ull << IDX_BITS) - 1;
// Largest rank of a core step.  Ranks up to LDSR = 32 keep the unit-triangular factor L[rows] in a static LDS array next to the
// matrix; a step with a larger rank (an elevated cross rank, approx_args_set_crossrank: 40 is what car7d 41^7 needs for a 1e-3
// value-iteration step) never fits LDS with its matrix anyway (41 x 40 x 40 doubles = 525 KB): it runs on global scratch and takes
// its MAXR x MAXR factor from the dynamic LDS block the matrix does not use.
constexpr int MAXR = 48, LDSR = 32;
constexpr size_t LDS_CAP_BYTES = 132 * 1024;  // dynamic LDS of the general core step: 160 KB of a CU minus its 17 KB of static arrays

struct Strides { long long s[MAXD]; };

// ------------------------------------------------------------------------------------------------ fiber index list
// idx[(a + r0 b) d + m] = I_k[a][m] (m < k) | 0 (m = k) | J_k[b][m - k - 1] (m > k), written by ONE workgroup into the step's own
// list -- and compared with what is there: if the step already holds this very list with values computed in the current
// generation (same sweep, same buffer layout), *skip = 1 and the fiber kernel that follows returns at once; every node of the
// list would be a memo hit returning the stored value, so the cached values ARE what it would produce (same bits, no counter
// moves).  With unchanged index sets that is every right-to-left step and every step of a confirming iteration.
struct NextList {
    int32_t *idx;                 // [F][d] list of the step (cached across iterations)
    const int32_t *I, *J;         // its index sets
    int r0, r1, k, d;
    int *skip;                    // out: 1 = list unchanged and values valid
    unsigned long long *tag;      // in/out: generation the cached values belong to
    unsigned long long gen;       // current generation
    int enable;                   // 0: never skip (sharded steps: the all-gather cannot be skipped on one rank only)
};

template <int BT>
__device__ inline void write_list_and_flag(const NextList &L)
{
    // a thread per fiber f = a + r0 b (one division per fiber; an element loop with e / d, f % r0, f / r0 per ENTRY spent 20-25 us of a
    // rank-48 core step here: 16 k entries, three runtime divisions each, twice)
    const int tid = threadIdx.x, d = L.d, k = L.k, F = L.r0 * L.r1, nj = d - 1 - k;
    int differs = 0;
    for (int f = tid; f < F; f += BT) {
        const int b = f / L.r0, a = f - b * L.r0;
        const int32_t *Ia = L.I + a * k, *Jb = L.J + b * nj;
        const int32_t *row = L.idx + f * d;
        for (int mm = 0; mm < k; mm++) differs |= (row[mm] != Ia[mm]);
        differs |= (row[k] != 0);
        for (int mm = 0; mm < nj; mm++) differs |= (row[k + 1 + mm] != Jb[mm]);
    }
    const int any = __syncthreads_or(differs);
    const bool valid = L.enable && (*L.tag == L.gen) && !any;
    if (!valid)
        for (int f = tid; f < F; f += BT) {
            const int b = f / L.r0, a = f - b * L.r0;
            const int32_t *Ia = L.I + a * k, *Jb = L.J + b * nj;
            int32_t *row = L.idx + f * d;
            for (int mm = 0; mm < k; mm++) row[mm] = Ia[mm];
            row[k] = 0;
            for (int mm = 0; mm < nj; mm++) row[k + 1 + mm] = Jb[mm];
        }
    __syncthreads();
    if (tid == 0) { *L.skip = valid ? 1 : 0; *L.tag = L.gen; }
}

__global__ void __launch_bounds__(256) k_cross_idx(const NextList L) { write_list_and_flag<256>(L); }

// ------------------------------------------------------------------------------------------------ node memo
// key word: [63:49] epoch of the sweep (never 0) | [48] pending | [47:0] node id.  A slot whose epoch is not the current one
// is free (the table is cleared by advancing the epoch).  Lookup and insertion in one pass: a hit takes the stored value
// (the reference's memo keeps the first value, bellman.c:1349-1353), a miss stores its own.  The same node twice in ONE
// batch means the same fiber twice (one varying dimension per batch), i.e. identical values: the loser of the race keeps
// its own.
constexpr unsigned long long EPOCH_MASK = MEMO_EPOCH_MASK, ID_MASK = MEMO_ID_MASK;

// growth: the entries of the current epoch move into a larger table (same probe rule; no concurrent lookups)
__global__ void k_cross_memo_rehash(const unsigned long long *__restrict__ okeys, const double *__restrict__ ovals, unsigned long long ocap,
                                    unsigned long long *keys, double *vals, unsigned long long capmask, int shift, unsigned long long epoch_bits)
{
    for (unsigned long long s = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; s < ocap; s += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long K = okeys[s];
        if ((K & EPOCH_MASK) != epoch_bits) continue;
        unsigned long long slot = ((K & ID_MASK) * 0x9E3779B97F4A7C15ull) >> shift;
        for (;; slot = (slot + 1) & capmask)
            if (atomicCAS(&keys[slot], 0ull, K) == 0ull) { vals[slot] = ovals[s]; break; }
    }
}

__global__ void k_cross_memo(const int32_t *__restrict__ idx, double *__restrict__ out, long total, int N, int d, int k, Strides S,
                             unsigned long long *keys, double *vals, unsigned long long capmask, int shift, unsigned long long epoch_bits,
                             unsigned long long *counters)
{ // the memo as a pass of its own, behind Bellman kernels that do not carry it in their epilogue (kernel_common.hpp: memo_merge)
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int inserted = 0, overflow = 0;
    if (e < total) {
        const long f = e / N;
        const int j = (int)(e - f * N);
        unsigned long long id = (unsigned long long)j * (unsigned long long)S.s[k];
        for (int m = 0; m < d; m++)
            if (m != k) id += (unsigned long long)idx[f * d + m] * (unsigned long long)S.s[m];
        const double v = out[e];
        if (v != v) overflow = 2; // a NaN row: the mark of a rank whose fibers failed in a sharded step (step_fibers) -- every rank sees it here
        else out[e] = memo_merge(keys, vals, capmask, shift, epoch_bits, id, v, inserted, overflow);
    }
    const unsigned long long mask = __ballot(inserted);
    if (mask != 0 && (threadIdx.x & 63) == (unsigned)__ffsll((long long)mask) - 1) atomicAdd(&counters[0], (unsigned long long)__popcll(mask));
    if (overflow) atomicMax(&counters[3], (unsigned long long)overflow); // 1: memo full, 2: a peer's rows arrived as NaN
}

// ------------------------------------------------------------------------------------------------ core step
struct CoreArgs {
    const double *out; // [r0 r1][N] fiber values, fiber f = a + r0 b
    int r0, N, r1, k, d;
    int dir;       // 0: left-to-right (rows (a, j) = a + r0 j, columns b); 1: right-to-left (rows (j, b) = j + N b, columns a)
    int copy_only; // last core of a half sweep: the fiber values are the core
    const int32_t *set_in; // dir 0: I_k [r0][k]; dir 1: J_k [r1][d-1-k]
    int32_t *set_out;      // dir 0: I_{k+1} [r1][k+1]; dir 1: J_{k-1} [r0][d-k]; read first: the rows it names start the pivot search
    double *G;             // core in the working layout G[a + r0 (j + N b)]
    double *work;          // m x n scratch in global memory when the matrix does not fit LDS
    unsigned long long *counters; // [1] rank-deficient factorisation seen, [2] maxvol swaps
    int warm;              // pivot search starts from the rows of the previous index set (c3sc_cross.c: WARM_BOOST_LOG2)
    double swap_tol;       // maxvol swaps while max |B| > 1 + swap_tol
    // the fiber index list of the NEXT core step, written at the end (its index sets are final then); next.idx == null: none
    NextList next;
    // confirm mode (k_cross_confirm): nothing is overwritten -- the index set the step produces is COMPARED with set_out, a
    // difference raises *mismatch; the core is written only by the right-to-left steps (they own G in a finished iteration)
    int confirm;
#ifdef C3SC_CORE_STAMPS
    unsigned long long *stamps; // diagnostic build: [8] cycle counts per phase of this step
#endif
    int *mismatch;
};

constexpr int WARM_BOOST_LOG2 = 6; // as in c3sc_cross.c

__device__ inline unsigned long long pivot_key(double x, unsigned long long index)
{ // larger |x| first (22 mantissa bits dropped: values equal to ~2e-10 relative tie), then the LOWER index
    const unsigned long long bits = (unsigned long long)__double_as_longlong(fabs(x));
    return ((bits >> IDX_BITS) << IDX_BITS) | (IDX_MASK - index);
}

// maximum of a u64 over the wavefront, in lane 63: DPP row shifts inside the rows of 16 lanes, then the two row broadcasts
// (six dependent VALU steps; the ds_bpermute butterfly it replaces was twelve LDS round trips on the critical path of every
// elimination column).  0 is the identity: lanes a shift does not reach read 0.
template <int CTRL, int ROWMASK>
__device__ inline unsigned long long dpp_u64(unsigned long long v)
{
    int lo = (int)(unsigned)(v & 0xffffffffull), hi = (int)(unsigned)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}
__device__ inline unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

__device__ inline unsigned long long wave_max_u64(unsigned long long v)
{
    v = umax64(v, dpp_u64<0x111, 0xf>(v)); // row_shr:1
    v = umax64(v, dpp_u64<0x112, 0xf>(v)); // row_shr:2
    v = umax64(v, dpp_u64<0x114, 0xf>(v)); // row_shr:4
    v = umax64(v, dpp_u64<0x118, 0xf>(v)); // row_shr:8   -> lane 15 of every row holds the row's maximum
    v = umax64(v, dpp_u64<0x142, 0xa>(v)); // row_bcast:15 into rows 1 and 3
    v = umax64(v, dpp_u64<0x143, 0xc>(v)); // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wavefront's maximum
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ inline unsigned long long block_max(unsigned long long v, unsigned long long *red /* [2][NT / 64] */, int &parity)
{
    v = wave_max_u64(v);
    unsigned long long *buf = red + parity * (NT / 64);
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long r = buf[0];
#pragma unroll
    for (int w = 1; w < NT / 64; w++) r = buf[w] > r ? buf[w] : r;
    parity ^= 1; // the next reduction writes the other buffer: one barrier per reduction is enough
    return r;
}

// rows named by the previous index set (P.set_out before it is overwritten), matched through the tuples of P.set_in; both
// sets are first brought into LDS in one coalesced round trip (matching them in global memory was ~10 dependent loads)
__device__ inline void mark_warm_rows(const CoreArgs &P, int m, int n, unsigned char *warmf, int *s_in, int *s_old)
{
    const int tid = threadIdx.x;
    const int lin = P.dir == 0 ? P.k : P.d - 1 - P.k, nin = P.dir == 0 ? P.r0 : P.r1; // set_in: nin tuples of length lin
    const int lold = lin + 1;                                                           // set_out: n tuples of length lin + 1
    for (int i = tid; i < m; i += NT) warmf[i] = 0;
    if (P.warm) {
        for (int e = tid; e < nin * lin; e += NT) s_in[e] = P.set_in[e];
        for (int e = tid; e < n * lold; e += NT) s_old[e] = P.set_out[e];
    }
    __syncthreads();
    // one thread per (old tuple q, candidate tuple a): the first matching a (lowest index, as a sequential scan would find it) names
    // the row.  The sequential scan it replaces -- one thread per old tuple, up to nin x lin dependent LDS reads -- was 6-10 k
    // cycles of a 64 k-cycle step.
    __shared__ int firsta[MAXR];
    if (tid < MAXR) firsta[tid] = 0x7fffffff;
    __syncthreads();
    if (P.warm)
        for (int e = tid; e < n * nin; e += NT) {
            const int q = e / nin, a = e % nin;
            const int *u = s_old + q * lold;
            const int off = P.dir == 0 ? 0 : 1; // dir 0: (u_0..u_{k-1}, j) against I_k[a]; dir 1: (j, v_1..) against J_k[a]
            bool eq = true;
            for (int t = 0; t < lin; t++) eq = eq && (s_in[a * lin + t] == u[off + t]);
            if (eq) atomicMin(&firsta[q], a);
        }
    __syncthreads();
    if (P.warm && tid < n && firsta[tid] != 0x7fffffff) {
        const int *u = s_old + tid * lold;
        const int a = firsta[tid];
        if (P.dir == 0) { // row a + r0 j
            const int j = u[lin];
            if (j >= 0 && j < P.N) warmf[a + P.r0 * j] = 1;
        } else { // row j + N b
            const int j = u[0];
            if (j >= 0 && j < P.N) warmf[j + P.N * a] = 1;
        }
    }
    __syncthreads();
}

// canonical order of the result: rows ascending; pos[c] = position of column c of B in the output (n <= 32: rank counting)
__device__ inline void sort_rows(int n, const int *rows, int *srows, int *pos)
{
    const int tid = threadIdx.x;
    if (tid < n) {
        const int r = rows[tid];
        int rank = 0;
        for (int q = 0; q < n; q++) rank += (rows[q] < r) || (rows[q] == r && q < tid);
        srows[rank] = r;
        pos[tid] = rank;
    }
    __syncthreads();
}

__device__ inline void write_sets_and_next(const CoreArgs &P, int n, const int *srows, const double *pivabs, int nswaps)
{
    const int tid = threadIdx.x, r0 = P.r0, N = P.N;
    if (!P.copy_only) {
        int differs = 0;
        if (P.dir == 0) {
            const int k = P.k;
            for (int e = tid; e < n * (k + 1); e += NT) {
                const int q = e / (k + 1), t = e % (k + 1);
                const int row = srows[q], a = row % r0, j = row / r0;
                const int v = t < k ? P.set_in[a * k + t] : j;
                if (P.confirm) differs |= (P.set_out[e] != v);
                else P.set_out[e] = v;
            }
        } else {
            const int len = P.d - P.k;
            for (int e = tid; e < n * len; e += NT) {
                const int q = e / len, t = e % len;
                const int row = srows[q], j = row % N, b = row / N;
                const int v = t == 0 ? j : P.set_in[b * (len - 1) + (t - 1)];
                if (P.confirm) differs |= (P.set_out[e] != v);
                else P.set_out[e] = v;
            }
        }
        if (P.confirm && differs) atomicOr(P.mismatch, 1);
        if (tid == 0) {
            double mx = 0.0, mn = INFINITY;
            for (int c = 0; c < n; c++) { mx = pivabs[c] > mx ? pivabs[c] : mx; mn = pivabs[c] < mn ? pivabs[c] : mn; }
            if (!(mn > 1e-12 * mx)) atomicExch(&P.counters[1], 1ull);
            if (nswaps) atomicAdd(&P.counters[2], (unsigned long long)nswaps);
        }
    }
    if (P.next.idx) { // fiber list of the next core step (k_cross_idx's job, without a launch of its own)
        __syncthreads();
        write_list_and_flag<NT>(P.next);
    }
}

// The matrix lives in LDS (column-major m x n; in global scratch when it does not fit), thread t owns rows t, t + NT, ...
// Everything a thread does to one of its rows is written as chunks of CH columns: the loads of a chunk are issued together
// and the arithmetic follows (left to itself the compiler keeps one LDS round trip per element in flight, because every
// store into the matrix might alias the next load: the kernel was 45 us of pure latency that way).
constexpr int CH_LDS = 8, CH_GLOBAL = 8; // 24 loads per trip on global scratch were measured SLOWER (rank-40 steps: 16.6 -> 21.4 ms per sweep: the
                                        // chunk is issued in full even where few trailing columns remain)

#ifdef C3SC_CORE_STAMPS
#define LU_STAMP(slot) do { if (threadIdx.x == 0 && P.stamps) { const unsigned long long now__ = clock64(); P.stamps[slot] += now__ - tlu__; tlu__ = now__; } } while (0)
#define CORE_STAMP(slot) do { __syncthreads(); if (threadIdx.x == 0 && P.stamps) { const unsigned long long now__ = clock64(); P.stamps[slot] += now__ - tlast__; tlast__ = now__; } } while (0)
#else
#define CORE_STAMP(slot) do { } while (0)
#define LU_STAMP(slot) do { } while (0)
#endif
template <int NRS, int QG>
__device__ __forceinline__ void core_step_regs(const CoreArgs &P, int m, int n, double *Lr, int LS, double *rowv, double *pivabs, int *rows, int *srows,
                                               int *pos, unsigned long long *red, unsigned char *warmf, int *s_in, int *s_old);

template <bool INLDS>
__device__ __forceinline__ void core_step(const CoreArgs &P)
{
#pragma clang fp contract(off) // products and sums round separately, as in the host twin (ISO C): the two return the same bits
#ifdef C3SC_CORE_STAMPS
    unsigned long long tlast__ = clock64();
#endif
    constexpr int CH = INLDS ? CH_LDS : CH_GLOBAL;
    extern __shared__ double smem[];
    __shared__ double Lr_s[LDSR * LDSR];
    double *Lr = INLDS ? Lr_s : smem; // !INLDS: MAXR x MAXR doubles of dynamic LDS (the matrix lives in global scratch)
    constexpr int LS = INLDS ? LDSR : MAXR;
    __shared__ double rowv[MAXR];
    __shared__ double pivabs[MAXR];
    __shared__ int rows[MAXR], srows[MAXR], pos[MAXR];
    __shared__ unsigned long long red[2 * (NT / 64)];
    __shared__ unsigned char warmf[MAXROWS];
    __shared__ int s_in[MAXR * MAXD], s_old[MAXR * MAXD];
    const int tid = threadIdx.x;
    const int r0 = P.r0, N = P.N, r1 = P.r1;
    if (P.copy_only) { // G[a + r0 (j + N b)] = out[(a + r0 b) N + j]
        if (P.dir == 0) { if (!P.confirm) write_sets_and_next(P, 0, srows, pivabs, 0); return; } // the last left-to-right step: its core (the fiber values) is rewritten by the first right-to-left step
        const int total = r0 * N * r1;
        for (int e = tid; e < total; e += NT) {
            const int a = e % r0, j = (e / r0) % N, b = e / (r0 * N);
            P.G[e] = P.out[(a + r0 * b) * N + j];
        }
        write_sets_and_next(P, 0, srows, pivabs, 0);
        return;
    }
    const int m = P.dir == 0 ? r0 * N : N * r1, n = P.dir == 0 ? r1 : r0;
    if (INLDS && n <= 16 && m <= 2 * NT) { // one or two rows per thread, whole rows in registers (car7d at the rank cap: 410 x 10; dubins3d on 101 nodes: 808 x 8)
        if (m <= NT) core_step_regs<16, 1>(P, m, n, Lr, LS, rowv, pivabs, rows, srows, pos, red, warmf, s_in, s_old);
        else core_step_regs<16, 2>(P, m, n, Lr, LS, rowv, pivabs, rows, srows, pos, red, warmf, s_in, s_old);
        return;
    }
    double *A = INLDS ? smem : P.work;
    for (int i = tid; i < m; i += NT) { // load the fiber matrix, a row per thread
        int base, step;
        if (P.dir == 0) { base = (i % r0) * N + i / r0; step = r0 * N; } // out[(a + r0 c) N + j], i = a + r0 j
        else { base = r0 * (i / N) * N + i % N; step = N; }              // out[(c + r0 b) N + j], i = j + N b
        for (int c0 = 0; c0 < n; c0 += CH) {
            double x[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) x[u] = (c0 + u < n) ? P.out[base + step * (c0 + u)] : 0.0;
#pragma unroll
            for (int u = 0; u < CH; u++) if (c0 + u < n) A[i + (c0 + u) * m] = x[u];
        }
    }
    CORE_STAMP(0);
    mark_warm_rows(P, m, n, warmf, s_in, s_old);
    CORE_STAMP(1);
    const double boost = (double)(1 << WARM_BOOST_LOG2);
    int parity = 0;
    unsigned used = 0; // bit q: my q-th row (row tid + q NT) is a pivot row
    // ---- tall LU with row pivoting: A = P L U; afterwards A holds L below the pivots (multipliers in place)
    for (int kc = 0; kc < n; kc++) {
        unsigned long long key = 0;
        for (int i = tid, q = 0; i < m; i += NT, q++)
            if (!((used >> q) & 1u)) {
                const double x = A[i + kc * m];
                const unsigned long long kk = pivot_key(warmf[i] ? x * boost : x, (unsigned long long)i);
                key = kk > key ? kk : key;
            }
        const unsigned long long best = block_max(key, red, parity);
        const int p = (int)(IDX_MASK - (best & IDX_MASK));
        const double dp = A[p + kc * m];
        if (tid == 0) { rows[kc] = p; pivabs[kc] = fabs(dp); }
        if (p % NT == tid) used |= 1u << (p / NT);
        const double inv = dp != 0.0 ? 1.0 / dp : 0.0;
        for (int i = tid, q = 0; i < m; i += NT, q++) {
            if ((used >> q) & 1u) continue;
            const double l = A[i + kc * m] * inv;
            A[i + kc * m] = l;
            for (int c0 = kc + 1; c0 < n; c0 += CH) { // every row, also with a zero multiplier (as lu_maxvol: no skipping, no sign of zero to keep)
                double x[CH], pc[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const int c = c0 + u < n ? c0 + u : n - 1;
                    x[u] = A[i + c * m];
                    pc[u] = A[p + c * m];
                }
#pragma unroll
                for (int u = 0; u < CH; u++) x[u] -= l * pc[u];
#pragma unroll
                for (int u = 0; u < CH; u++) if (c0 + u < n) A[i + (c0 + u) * m] = x[u];
            }
        }
        // no barrier here: the next column's pivot search reads a thread's own rows only, and the barrier inside its block_max
        // stands between this update of a row and any other thread reading it as the next pivot row (pivot rows are never
        // written again)
    }
    __syncthreads();
    CORE_STAMP(2);
    // ---- B = L inv(L[rows]): L[rows] is unit lower triangular in pivot order
    for (int e = tid; e < n * n; e += NT) {
        const int q = e / n, j = e % n;
        Lr[q * LS + j] = j < q ? A[rows[q] + j * m] : (j == q ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int i = tid, q = 0; i < m; i += NT, q++) {
        if ((used >> q) & 1u) continue;
        for (int j = n - 1; j >= 0; j--) { // x Lr = l, in place: s = l_j - sum_{t > j} x_t Lr[t][j], t ascending
            double s = A[i + j * m];
            for (int t0 = j + 1; t0 < n; t0 += CH) {
                double xt[CH], w[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const int t = t0 + u < n ? t0 + u : n - 1;
                    xt[u] = A[i + t * m];
                    w[u] = Lr[t * LS + j];
                }
#pragma unroll
                for (int u = 0; u < CH; u++) if (t0 + u < n) s -= xt[u] * w[u];
            }
            A[i + j * m] = s;
        }
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += NT) { // pivot rows: unit vectors
        const int q = e / n, j = e % n;
        A[rows[q] + j * m] = (j == q) ? 1.0 : 0.0;
    }
    __syncthreads();
    CORE_STAMP(3);
    // ---- maxvol: swap rows until the largest entry of B is <= 1 + swap_tol
    int nswaps = 0;
    for (int it = 0; it < 200; it++) {
        unsigned long long key = 0;
        for (int i = tid; i < m; i += NT)
            for (int c0 = 0; c0 < n; c0 += CH) {
                double x[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) x[u] = A[i + (c0 + u < n ? c0 + u : n - 1) * m];
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const unsigned long long kk = (c0 + u < n) ? pivot_key(x[u], (unsigned long long)((c0 + u) * m + i)) : 0ull;
                    key = kk > key ? kk : key;
                }
            }
        const unsigned long long best = block_max(key, red, parity);
        const int lin = (int)(IDX_MASK - (best & IDX_MASK));
        const int bj = lin / m, bi = lin % m;
        const double piv = A[bi + bj * m];
        if (!(fabs(piv) > 1.0 + P.swap_tol)) break;
        if (tid < n) rowv[tid] = A[bi + tid * m] - (tid == bj ? 1.0 : 0.0);
        __syncthreads();
        for (int i = tid; i < m; i += NT) {
            const double cv = A[i + bj * m] / piv;
            for (int c0 = 0; c0 < n; c0 += CH) { // every row, zero multipliers too (as lu_maxvol)
                    double x[CH], rc[CH];
#pragma unroll
                    for (int u = 0; u < CH; u++) {
                        const int c = c0 + u < n ? c0 + u : n - 1;
                        x[u] = A[i + c * m];
                        rc[u] = rowv[c];
                    }
#pragma unroll
                    for (int u = 0; u < CH; u++) x[u] -= cv * rc[u];
#pragma unroll
                    for (int u = 0; u < CH; u++) if (c0 + u < n) A[i + (c0 + u) * m] = x[u];
                }
        }
        if (tid == 0) rows[bj] = bi;
        nswaps++;
        // no barrier here either: the next search reads own rows, its block_max barrier orders the rest (rowv is rewritten only
        // after that barrier, when every thread has finished this update)
    }
    __syncthreads();
    CORE_STAMP(4);
    sort_rows(n, rows, srows, pos);
    // ---- results, columns in the order of ascending rows.  A left-to-right step's core is never read: the right-to-left half of
    // the iteration rewrites every core (c3sc_cross.c frees the left-to-right train unseen) -- only its index set counts
    if (P.dir == 1)
    for (int i = tid; i < m; i += NT)
        for (int c0 = 0; c0 < n; c0 += CH) {
            double x[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) x[u] = A[i + (c0 + u < n ? c0 + u : n - 1) * m];
#pragma unroll
            for (int u = 0; u < CH; u++)
                if (c0 + u < n) {
                    P.G[pos[c0 + u] + r0 * i] = x[u]; // G[a' + r0 cc], cc = i
                }
        }
    CORE_STAMP(5);
    write_sets_and_next(P, n, srows, pivabs, nswaps);
    CORE_STAMP(6);
#ifdef C3SC_CORE_STAMPS
    if (threadIdx.x == 0 && P.stamps) P.stamps[7] += (unsigned long long)nswaps;
#endif
}

template <bool INLDS>
__global__ void __launch_bounds__(NT) k_cross_core(const CoreArgs P) { core_step<INLDS>(P); }

// ------------------------------------------------------------------------------ the core step of a matrix that does not fit LDS
// Ranks above 32 (an elevated cross rank: 1968 x 48 at car7d's 41 nodes) and long dimensions at rank > 16 keep the matrix in
// global scratch (L2-resident: 755 KB).  The LDS form above run on that scratch is a chain of dependent L2 round trips -- every
// chunk of a row's update is load -> arithmetic -> store, and the next chunk's loads wait for those stores because they might
// alias: 0.6-1.1 ms per rank-48 step, 17-23 ms of a 29 ms value-iteration sweep.  Same arithmetic, element by element and in the
// same order (so still the bit-for-bit twin of lu_maxvol, c3sc_cross.c), organised around what is cheap here:
//   * LU LEFT-looking: column kc of a row is brought up to date when it is needed, x = a - sum_{t < kc} l_t U[t][kc] (t ascending:
//     the order in which the right-looking form subtracts), from the row's own multipliers -- loads that depend on no store of
//     this column -- and from U, the pivot rows' updated entries, which live in LDS (MAXR x MAXR).  One store per row and column
//     instead of a read-modify-write of the whole trailing matrix.  The new pivot row's U entries are formed by ONE wavefront
//     (lane = column; the row's multipliers sit in the lanes of a register and are read with v_readlane), while the others wait.
//   * substitution, maxvol updates: a row at a time in REGISTERS (NR doubles, statically indexed, padded with +0 beyond n: a
//     product of two +0 subtracts as +0 and leaves every value, -0 included, as it is): one round trip per row.  The pivot search
//     of maxvol rides on the pass that produced the values (no separate read of the matrix).
// The pivot search of a thread over the entries of ONE row in ascending column order: the largest magnitude with 22 mantissa bits
// dropped, the first one among equals -- what the maximum of pivot_key(x, t m + i) over the row gives, at six instructions an
// entry instead of ten (the key itself is formed once per row, row_key).
struct RowBest { unsigned long long tm; int t; };
__device__ __forceinline__ void track(RowBest &b, double x, int t)
{
    const unsigned long long tm = ((unsigned long long)__double_as_longlong(fabs(x)) >> IDX_BITS) << IDX_BITS;
    const bool better = tm > b.tm;
    b.tm = better ? tm : b.tm;
    b.t = better ? t : b.t;
}
__device__ __forceinline__ unsigned long long row_key(const RowBest &b, int m, int i) { return b.tm | (IDX_MASK - (unsigned long long)(b.t * m + i)); }

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

template <int NR>
__device__ __forceinline__ void core_step_global(const CoreArgs &P)
{
#pragma clang fp contract(off)
#ifdef C3SC_CORE_STAMPS
    unsigned long long tlast__ = clock64();
#endif
    constexpr int CH = 8, LS = MAXR, QG = 4; // QG rows of a thread advance together through a column of the LU
    extern __shared__ double smem[];         // MAXR x MAXR: U[t][c] during the LU, then L[rows] transposed (LrT[j][t] = Lr[t][j])
    double *U = smem;
    __shared__ double rowv[MAXR];
    __shared__ double pivabs[MAXR];
    __shared__ int rows[MAXR], srows[MAXR], pos[MAXR];
    __shared__ unsigned long long red[2 * (NT / 64)];
    __shared__ unsigned char warmf[MAXROWS];
    __shared__ int s_in[MAXR * MAXD], s_old[MAXR * MAXD];
    const int tid = threadIdx.x;
    const int r0 = P.r0, N = P.N, r1 = P.r1;
    if (P.copy_only) { // G[a + r0 (j + N b)] = out[(a + r0 b) N + j]
        if (P.dir == 0) { if (!P.confirm) write_sets_and_next(P, 0, srows, pivabs, 0); return; } // the last left-to-right step: its core (the fiber values) is rewritten by the first right-to-left step
        const int total = r0 * N * r1;
        for (int e = tid; e < total; e += NT) {
            const int a = e % r0, j = (e / r0) % N, b = e / (r0 * N);
            P.G[e] = P.out[(a + r0 * b) * N + j];
        }
        write_sets_and_next(P, 0, srows, pivabs, 0);
        return;
    }
    const int m = P.dir == 0 ? r0 * N : N * r1, n = P.dir == 0 ? r1 : r0;
    double *A = P.work;
    // the fiber matrix, a row per thread: gathered loads (stride N or r0 N doubles, every line of `out` is used by neighbouring
    // rows and stays in the caches) and coalesced stores.  Measured the other way round -- out read in its own order, elements
    // scattered into A -- the scattered 8-byte stores cost twice as much (84 k -> 150-220 k cycles for 1968 x 48), with or
    // without the index divisions.
    for (int i = tid; i < m; i += NT) {
        int base, step;
        if (P.dir == 0) { base = (i % r0) * N + i / r0; step = r0 * N; } // out[(a + r0 c) N + j], i = a + r0 j
        else { base = r0 * (i / N) * N + i % N; step = N; }              // out[(c + r0 b) N + j], i = j + N b
        for (int c0 = 0; c0 < n; c0 += CH) {
            double x[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) x[u] = (c0 + u < n) ? P.out[base + step * (c0 + u)] : 0.0;
#pragma unroll
            for (int u = 0; u < CH; u++) if (c0 + u < n) A[i + (c0 + u) * m] = x[u];
        }
        for (int c = n; c < NR; c++) A[i + c * m] = 0.0; // the padding columns of the rows' register image
    }
    CORE_STAMP(0);
    mark_warm_rows(P, m, n, warmf, s_in, s_old);
    CORE_STAMP(1);
    const double boost = (double)(1 << WARM_BOOST_LOG2);
    int parity = 0;
    unsigned used = 0; // bit q: my q-th row (row tid + q NT) is a pivot row
    const int ngroups = (m + QG * NT - 1) / (QG * NT);
#ifdef C3SC_CORE_STAMPS
    unsigned long long tlu__ = clock64();
#endif
    // ---- tall LU with row pivoting.  Every element sees x -= l_t U[t][c] for t = 0, 1, 2, ... in this order, as in lu_maxvol.
    if (m <= QG * NT) {
        // Panels of PW columns, a thread's (at most QG) rows of the panel in REGISTERS:
        //   A  the panel is brought up to date through column c0 - 1 from the rows' stored multipliers (one load per row and t,
        //      used for PW columns) and U[t][c0 ..] (LDS, read once per thread for its QG rows);
        //   B  column by column inside the panel: pivot search on the registers, the pivot row's owner publishes its panel
        //      entries (they ARE U[kc][c0 ..]), everybody scales its entry (the multiplier, stored at once) and updates the
        //      rest of its panel registers: two barriers per column, no global round trip on the critical path;
        //   C  at the end of the panel the U entries of its PW pivot rows BEYOND the panel: wavefront w takes pivot row w (lane =
        //      column) through the terms t < c0, then one wavefront finishes the PW x PW triangle inside the panel.
        constexpr int PW = 8;
        __shared__ double part[PW * MAXR]; // [w][c]: pivot row w of the panel through t < c0
        __shared__ double lin[PW * PW];    // [w][tt]: multiplier of pivot row w for column c0 + tt (tt < w)
        __shared__ double slots[2 * (NT / 64) * PW]; // [parity][wavefront][v]: the wavefront's candidate row, panel columns
        static_assert(QG == 4, "the slot hand-over below names the rows one by one");
        int row[QG];
        unsigned rowc[QG], warmbits = 0; // bit qq: row qq carries the warm-start boost
#pragma unroll
        for (int qq = 0; qq < QG; qq++) {
            row[qq] = tid + qq * NT;
            rowc[qq] = row[qq] < m ? (unsigned)row[qq] : 0u; // a thread without a row there reads row 0 and throws the result away
            warmbits |= (row[qq] < m && warmf[rowc[qq]]) ? (1u << qq) : 0u;
        }
        for (int c0 = 0; c0 < n; c0 += PW) {
            const int np = n - c0 < PW ? n - c0 : PW;
            double acc[QG][PW];
#pragma unroll
            for (int v = 0; v < PW; v++) {
                const unsigned off = (unsigned)((c0 + v < NR ? c0 + v : NR - 1) * m);
#pragma unroll
                for (int qq = 0; qq < QG; qq++) acc[qq][v] = A[rowc[qq] + off];
            }
            // ---- A: terms t < c0 (c0 is a multiple of PW: whole chunks of CA)
            {
                constexpr int CA = 4;
                auto fetch = [&](int t0, double (&l)[QG][CA]) {
#pragma unroll
                    for (int u = 0; u < CA; u++) {
                        const unsigned off = (unsigned)((t0 + u) * m);
#pragma unroll
                        for (int qq = 0; qq < QG; qq++) l[qq][u] = A[rowc[qq] + off];
                    }
                };
                auto apply = [&](int t0, const double (&l)[QG][CA]) {
#pragma unroll
                    for (int u = 0; u < CA; u++) {
                        double w[PW];
#pragma unroll
                        for (int v = 0; v < PW; v++) w[v] = U[(t0 + u) * LS + (c0 + v < MAXR ? c0 + v : MAXR - 1)];
#pragma unroll
                        for (int qq = 0; qq < QG; qq++)
#pragma unroll
                            for (int v = 0; v < PW; v++) acc[qq][v] -= l[qq][u] * w[v];
                    }
                };
                double la[QG][CA], lb[QG][CA]; // two chunks: the next one is on its way while this one is applied
                if (c0 > 0) fetch(0, la);
                for (int t0 = 0; t0 < c0; t0 += 2 * CA) {
                    const bool second = t0 + CA < c0;
                    if (second) fetch(t0 + CA, lb);
                    apply(t0, la);
                    if (second) {
                        if (t0 + 2 * CA < c0) fetch(t0 + 2 * CA, la);
                        apply(t0 + CA, lb);
                    }
                }
            }
            LU_STAMP(8);
            // ---- B: the panel's columns.  One barrier per column: every wavefront leaves its best candidate's key AND that row's panel
            // registers in its slot; behind the barrier everybody reads the winning slot -- it is U[kc][kc .. c0 + PW - 1].
            static_for<0, PW>([&](auto uu) {
                constexpr int u = decltype(uu)::value;
                if (u < np) {
                    const int kc = c0 + u;
                    unsigned long long key = 0;
                    unsigned kbit = 0;
#pragma unroll
                    for (int qq = 0; qq < QG; qq++) {
                        const double x = acc[qq][u];
                        const unsigned long long kk = (row[qq] < m && !((used >> qq) & 1u)) ? pivot_key(((warmbits >> qq) & 1u) ? x * boost : x, (unsigned long long)row[qq]) : 0ull;
                        const bool better = kk > key;
                        key = better ? kk : key;
                        kbit = better ? (1u << qq) : kbit;
                    }
                    const unsigned long long wmax = wave_max_u64(key);
                    double *slot = slots + (parity * (NT / 64) + (tid >> 6)) * PW;
                    if (key == wmax && key != 0) { // one lane (the index is part of the key) leaves its best row's panel registers: a branch
                        // per row -- indexing acc by a run-time row (or a chain of selects, which the compiler turns into that) sends acc to scratch
                        if (kbit == 1u) {
#pragma unroll
                            for (int v = u; v < PW; v++) { double t = acc[0][v]; asm volatile("" : "+v"(t)); slot[v] = t; } // (opaque: the four branches must not be merged into one indexed read)
                        } else if (kbit == 2u) {
#pragma unroll
                            for (int v = u; v < PW; v++) { double t = acc[1][v]; asm volatile("" : "+v"(t)); slot[v] = t; } // (opaque: the four branches must not be merged into one indexed read)
                        } else if (kbit == 4u) {
#pragma unroll
                            for (int v = u; v < PW; v++) { double t = acc[2][v]; asm volatile("" : "+v"(t)); slot[v] = t; } // (opaque: the four branches must not be merged into one indexed read)
                        } else {
#pragma unroll
                            for (int v = u; v < PW; v++) { double t = acc[3][v]; asm volatile("" : "+v"(t)); slot[v] = t; } // (opaque: the four branches must not be merged into one indexed read)
                        }
                    }
                    if ((tid & 63) == 0) red[parity * (NT / 64) + (tid >> 6)] = wmax;
                    __syncthreads();
                    unsigned long long best = red[parity * (NT / 64)];
                    int wb = 0;
#pragma unroll
                    for (int w2 = 1; w2 < NT / 64; w2++) {
                        const unsigned long long r = red[parity * (NT / 64) + w2];
                        if (r > best) { best = r; wb = w2; }
                    }
                    const double *win = slots + (parity * (NT / 64) + wb) * PW;
                    parity ^= 1; // the next column writes the other set of slots: one barrier per column is enough
                    const int p = (int)(IDX_MASK - (best & IDX_MASK));
                    if (key == best && key != 0) used |= kbit; // the pivot row's owner
                    const double dp = win[u];
                    if (tid == 0) { rows[kc] = p; pivabs[kc] = fabs(dp); }
                    const double inv = dp != 0.0 ? 1.0 / dp : 0.0;
                    double w[PW];
#pragma unroll
                    for (int v = u + 1; v < PW; v++) w[v] = win[v];
#pragma unroll
                    for (int qq = 0; qq < QG; qq++)
                        if (row[qq] < m && !((used >> qq) & 1u)) {
                            const double l = acc[qq][u] * inv;
                            A[row[qq] + kc * m] = l;
#pragma unroll
                            for (int v = u + 1; v < PW; v++) acc[qq][v] -= l * w[v];
                        }
                }
            });
            LU_STAMP(9);
            // ---- C: U[c0 + w][c] for c >= c0 + PW
            if (c0 + PW < n) {
                __syncthreads(); // the multipliers of the panel's last column are stored (the pivot rows' are read below)
                const int wv = tid >> 6, lane = tid & 63;
                if (wv < PW) { // np = PW here
                    const int pw = rows[c0 + wv];
                    const int c = lane < n ? lane : n - 1;
                    double a0 = A[pw + c * m];
                    for (int tb = 0; tb < c0; tb += 64) { // the pivot row's multipliers, 64 at a time in the lanes of a register
                        const double lpv = tb + lane < c0 ? A[pw + (tb + lane) * m] : 0.0;
                        const int lplo = (int)(unsigned)((unsigned long long)__double_as_longlong(lpv) & 0xffffffffull);
                        const int lphi = (int)(unsigned)((unsigned long long)__double_as_longlong(lpv) >> 32);
                        const int tn = c0 - tb < 64 ? c0 - tb : 64;
                        for (int t0 = 0; t0 < tn; t0 += CH) { // tn is a multiple of CH
                            double ut[CH];
#pragma unroll
                            for (int uq = 0; uq < CH; uq++) ut[uq] = U[(tb + t0 + uq) * LS + (c < MAXR ? c : MAXR - 1)];
#pragma unroll
                            for (int uq = 0; uq < CH; uq++) {
                                const unsigned lo = (unsigned)__builtin_amdgcn_readlane(lplo, t0 + uq), hi = (unsigned)__builtin_amdgcn_readlane(lphi, t0 + uq);
                                a0 -= __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)) * ut[uq];
                            }
                        }
                    }
                    if (lane >= c0 + PW && lane < n) part[wv * MAXR + lane] = a0;
                    if (lane < wv) lin[wv * PW + lane] = A[pw + (c0 + lane) * m];
                }
                __syncthreads();
                if (tid >= c0 + PW && tid < n) { // one wavefront (n <= 48): the triangle inside the panel, t = c0 .. c0 + w - 1 ascending
                    double av[PW];
#pragma unroll
                    for (int w2 = 0; w2 < PW; w2++) av[w2] = part[w2 * MAXR + tid];
#pragma unroll
                    for (int w2 = 0; w2 < PW; w2++) {
#pragma unroll
                        for (int tt = 0; tt < w2; tt++) av[w2] -= lin[w2 * PW + tt] * av[tt];
                        U[(c0 + w2) * LS + tid] = av[w2];
                    }
                }
                __syncthreads();
            }
            LU_STAMP(10);
        }
    } else {
        for (int kc = 0; kc < n; kc++) {
            unsigned long long key = 0;
            double xs[QG]; // the first group's entries of this column stay in registers until they are scaled
    #pragma unroll
            for (int qq = 0; qq < QG; qq++) xs[qq] = 0.0;
            for (int g = 0; g < ngroups; g++) {
                double s[QG];
                int row[QG];
                unsigned rowc[QG]; // the row every load goes to: a row without work reads row 0 and throws the sum away (no predicated loads)
                bool live[QG];
    #pragma unroll
                for (int qq = 0; qq < QG; qq++) {
                    const int q = g * QG + qq;
                    row[qq] = tid + q * NT;
                    live[qq] = row[qq] < m && !((used >> q) & 1u);
                    rowc[qq] = live[qq] ? (unsigned)row[qq] : 0u;
                    s[qq] = A[rowc[qq] + (unsigned)(kc * m)];
                }
                // the row's multipliers of columns t0 .. t0 + CH - 1 (clamped to kc - 1: the tail is loaded twice and not used)
                auto fetch = [&](int t0, double (&l)[QG][CH]) {
    #pragma unroll
                    for (int u = 0; u < CH; u++) {
                        const unsigned off = (unsigned)((t0 + u < kc ? t0 + u : kc - 1) * m);
    #pragma unroll
                        for (int qq = 0; qq < QG; qq++) l[qq][u] = A[rowc[qq] + off];
                    }
                };
                auto apply = [&](int t0, const double (&l)[QG][CH]) {
                    double w[CH];
    #pragma unroll
                    for (int u = 0; u < CH; u++) w[u] = U[(t0 + u < kc ? t0 + u : kc - 1) * LS + kc];
    #pragma unroll
                    for (int qq = 0; qq < QG; qq++)
    #pragma unroll
                        for (int u = 0; u < CH; u++)
                            if (t0 + u < kc) s[qq] -= l[qq][u] * w[u];
                };
                double la[QG][CH], lb[QG][CH]; // two chunks: the next one is on its way while this one is applied
                if (kc > 0) fetch(0, la);
                for (int t0 = 0; t0 < kc; t0 += 2 * CH) {
                    const bool second = t0 + CH < kc;
                    if (second) fetch(t0 + CH, lb);
                    apply(t0, la);
                    if (second) {
                        if (t0 + 2 * CH < kc) fetch(t0 + 2 * CH, la);
                        apply(t0 + CH, lb);
                    }
                }
    #pragma unroll
                for (int qq = 0; qq < QG; qq++)
                    if (live[qq]) {
                        if (g == 0) xs[qq] = s[qq];
                        else A[row[qq] + kc * m] = s[qq];
                        const unsigned long long kk = pivot_key(warmf[row[qq]] ? s[qq] * boost : s[qq], (unsigned long long)row[qq]);
                        key = kk > key ? kk : key;
                    }
            }
            LU_STAMP(8);
            const unsigned long long best = block_max(key, red, parity);
            LU_STAMP(9);
            const int p = (int)(IDX_MASK - (best & IDX_MASK));
            if (p % NT == tid) used |= 1u << (p / NT);
            if (tid < 64) { // U[kc][c], c >= kc: the pivot row's entries, up to date through column kc - 1 (lane = column)
                const int c = tid;
                double a0 = (c >= kc && c < n) ? A[p + c * m] : 0.0;
                const double a_in = a0;
                const double lpv = c < kc ? A[p + c * m] : 0.0; // lane t: the pivot row's multiplier of column t
                const int lplo = (int)(unsigned)((unsigned long long)__double_as_longlong(lpv) & 0xffffffffull);
                const int lphi = (int)(unsigned)((unsigned long long)__double_as_longlong(lpv) >> 32);
                const int cc = c < MAXR ? c : MAXR - 1;
                for (int t0 = 0; t0 < kc; t0 += CH) {
                    double ut[CH];
    #pragma unroll
                    for (int u = 0; u < CH; u++) ut[u] = U[(t0 + u < kc ? t0 + u : kc - 1) * LS + cc];
    #pragma unroll
                    for (int u = 0; u < CH; u++)
                        if (t0 + u < kc) {
                            const unsigned lo = (unsigned)__builtin_amdgcn_readlane(lplo, t0 + u), hi = (unsigned)__builtin_amdgcn_readlane(lphi, t0 + u);
                            const double lt = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                            a0 -= lt * ut[u];
                        }
                }
                // a row beyond the first QG of its thread has its entry of column kc up to date in the matrix already (it was stored
                // for the scaling below): nothing to subtract there
                if (c == kc && p >= QG * NT) a0 = a_in;
                if (c >= kc && c < n) U[kc * LS + c] = a0;
            }
            __syncthreads();
            LU_STAMP(10);
            const double dp = U[kc * LS + kc];
            if (tid == 0) { rows[kc] = p; pivabs[kc] = fabs(dp); }
            const double inv = dp != 0.0 ? 1.0 / dp : 0.0;
            for (int g = 0; g < ngroups; g++)
    #pragma unroll
                for (int qq = 0; qq < QG; qq++) {
                    const int q = g * QG + qq, i = tid + q * NT;
                    if (i < m && !((used >> q) & 1u)) A[i + kc * m] = (g == 0 ? xs[qq] : A[i + kc * m]) * inv;
                }
            LU_STAMP(11);
            // no barrier here: the next column reads a thread's own rows and U rows that were complete at the barrier above; the next
            // pivot row's multipliers are read behind the barrier inside the next block_max
        }
    }
    __syncthreads();
    CORE_STAMP(2);
    // ---- B = L inv(L[rows]): x Lr = l per non-pivot row, the row in registers
    double *LrT = smem; // LrT[j LS + t] = Lr[t][j] = A[rows[t] + j m] (j < t < n), +0 elsewhere
    for (int e = tid; e < NR * NR; e += NT) {
        const int j = e / NR, t = e % NR;
        LrT[j * LS + t] = (j < t && t < n) ? A[rows[t] + j * m] : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += NT) { // pivot rows: unit vectors (their multipliers are in LrT now)
        const int q = e / n, j = e % n;
        A[rows[q] + j * m] = (j == q) ? 1.0 : 0.0;
    }
    unsigned long long key = 0; // maxvol's first search rides on this pass (the unit rows cannot win it with an entry above 1 + swap_tol)
    for (int i = tid, q = 0; i < m; i += NT, q++) {
        if ((used >> q) & 1u) continue;
        double x[NR];
        int ms = m;
        asm volatile("" : "+s"(ms)); // the column offsets t m are formed here (scalar unit): hoisted out of the row loop they are 48 live registers, spilled
#pragma unroll
        for (int t = 0; t < NR; t++) x[t] = (A + t * ms)[i];
        int opaque = 0;
        asm volatile("" : "+v"(opaque)); // the factor is read where it is used: hoisted out of the row loop it is 1 128 live values
        const double *Lq = LrT + opaque; // (an offset, not the pointer: a laundered pointer loses its address space and reads FLAT)
        static_for<0, NR>([&](auto jj) { // s = l_j - sum_{t > j} x_t Lr[t][j], t ascending; j = NR - 1 .. 0
            constexpr int j = NR - 1 - decltype(jj)::value;
            double s = x[j];
            static_for<0, (NR - 1 - j + CH - 1) / CH>([&](auto cc) { // CH entries of the factor in flight (left alone the scheduler asks
                constexpr int t0 = j + 1 + decltype(cc)::value * CH; // for whole columns at once and spills)
                double w[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) if (t0 + u < NR) w[u] = Lq[j * LS + t0 + u];
#pragma unroll
                for (int u = 0; u < CH; u++) if (t0 + u < NR) s -= x[t0 + u] * w[u];
                __builtin_amdgcn_sched_barrier(0);
            });
            x[j] = s;
        });
        RowBest rb = {0ull, 0};
#pragma unroll
        for (int t = 0; t < NR; t++) { // the padding columns stay +0: they lose against any entry that is not zero (and come last among zeros)
            (A + t * ms)[i] = x[t];
            track(rb, x[t], t);
        }
        const unsigned long long kk = row_key(rb, m, i);
        key = kk > key ? kk : key;
    }
    CORE_STAMP(3);
    // ---- maxvol: swap rows until the largest entry of B is <= 1 + swap_tol
    int nswaps = 0;
    for (int it = 0; it < 200; it++) {
        const unsigned long long best = block_max(key, red, parity);
        if (best == 0) break; // m = n: every row is a pivot row, B is the identity and the first search saw no entry at all
        const int lin = (int)(IDX_MASK - (best & IDX_MASK));
        const int bj = lin / m, bi = lin % m;
        const double piv = A[bi + bj * m];
        if (!(fabs(piv) > 1.0 + P.swap_tol)) break;
        if (tid < MAXR) rowv[tid] = tid < n ? A[bi + tid * m] - (tid == bj ? 1.0 : 0.0) : 0.0;
        __syncthreads();
        key = 0;
        for (int i = tid; i < m; i += NT) { // the row in registers: updated and searched in one pass (zero multipliers are not skipped: as lu_maxvol)
            double x[NR];
            const double cv = A[i + bj * m] / piv;
            int ms = m;
            asm volatile("" : "+s"(ms));
#pragma unroll
            for (int t = 0; t < NR; t++) x[t] = (A + t * ms)[i];
            {
                int opaque = 0;
                asm volatile("" : "+v"(opaque));
                const double *rv = rowv + opaque;
#pragma unroll
                for (int t = 0; t < NR; t++) x[t] -= cv * rv[t]; // rowv is +0 beyond n: the padding stays +0
            }
            RowBest rb = {0ull, 0};
#pragma unroll
            for (int t = 0; t < NR; t++) {
                (A + t * ms)[i] = x[t];
                track(rb, x[t], t);
            }
            const unsigned long long kk = row_key(rb, m, i);
            key = kk > key ? kk : key;
        }
        if (tid == 0) rows[bj] = bi;
        nswaps++;
        // the next block_max's barrier stands between these stores and the reads of piv and rowv
    }
    __syncthreads();
    CORE_STAMP(4);
    sort_rows(n, rows, srows, pos);
    if (P.dir == 1) // (a left-to-right step's core is never read: see core_step)
    for (int i = tid; i < m; i += NT) // results, columns in the order of ascending rows
        for (int c0 = 0; c0 < n; c0 += CH) {
            double x[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) x[u] = A[i + (c0 + u < n ? c0 + u : n - 1) * m];
#pragma unroll
            for (int u = 0; u < CH; u++)
                if (c0 + u < n) {
                    P.G[pos[c0 + u] + r0 * i] = x[u];
                }
        }
    CORE_STAMP(5);
    write_sets_and_next(P, n, srows, pivabs, nswaps);
    CORE_STAMP(6);
#ifdef C3SC_CORE_STAMPS
    if (threadIdx.x == 0 && P.stamps) P.stamps[7] += (unsigned long long)nswaps;
#endif
}

template <int NR>
__global__ void __launch_bounds__(NT) k_cross_core_g(const CoreArgs P) { core_step_global<NR>(P); }

// ------------------------------------------------------------------------------ the core step of a small matrix, in registers
// At most QG NT rows (QG = 1 or 2 rows per thread) and NRS = 16 columns (car7d at its rank cap: 410 x 10): the WHOLE rows in
// registers from the gather of the fiber values to the write of the core -- no matrix in LDS at all.  The LU is the column phase of
// core_step_global with one panel (one barrier per column: every wavefront leaves its candidate's key and row in a slot, the
// winner's slot is the pivot row), the pivot rows' multipliers go to L[rows] in LDS as they are chosen, the substitution runs on the
// registers, and a maxvol swap is a broadcast of the swapped-in row through LDS and 2 n flops per row.  Same arithmetic in the same
// order as core_step and lu_maxvol: the same bits.  (Which of a thread's rows is meant is always decided by BRANCHES over opaque
// copies: a run-time index into the register rows, or a chain of selects the compiler turns into one, sends them to scratch.)
template <int NRS, int QG>
__device__ __forceinline__ void core_step_regs(const CoreArgs &P, int m, int n, double *Lr, int LS, double *rowv, double *pivabs, int *rows, int *srows,
                                               int *pos, unsigned long long *red, unsigned char *warmf, int *s_in, int *s_old)
{
#pragma clang fp contract(off)
#ifdef C3SC_CORE_STAMPS
    unsigned long long tlast__ = clock64();
#endif
    __shared__ double slots[2 * (NT / 64) * NRS];
    const int tid = threadIdx.x, r0 = P.r0, N = P.N;
    bool has[QG];
    int row[QG];
    double x[QG][NRS];
#pragma unroll
    for (int qq = 0; qq < QG; qq++) {
        row[qq] = tid + qq * NT;
        has[qq] = row[qq] < m;
        const int i = has[qq] ? row[qq] : 0;
        int base, step;
        if (P.dir == 0) { base = (i % r0) * N + i / r0; step = r0 * N; } // out[(a + r0 c) N + j], i = a + r0 j
        else { base = r0 * (i / N) * N + i % N; step = N; }              // out[(c + r0 b) N + j], i = j + N b
#pragma unroll
        for (int c = 0; c < NRS; c++) x[qq][c] = c < n ? P.out[base + step * c] : 0.0; // +0 beyond n: stays +0 through every update
    }
    for (int e = tid; e < NRS * NRS; e += NT) Lr[(e / NRS) * LS + e % NRS] = 0.0; // L[rows]: rows filled as the pivots are chosen, +0 elsewhere
    CORE_STAMP(0);
    mark_warm_rows(P, m, n, warmf, s_in, s_old);
    CORE_STAMP(1);
    unsigned warmbits = 0, used = 0;
    int mypos[QG];
#pragma unroll
    for (int qq = 0; qq < QG; qq++) { warmbits |= (has[qq] && warmf[has[qq] ? row[qq] : 0]) ? (1u << qq) : 0u; mypos[qq] = -1; }
    const double boost = (double)(1 << WARM_BOOST_LOG2);
    int parity = 0;
    // ---- LU: one barrier per column
    static_for<0, NRS>([&](auto uu) {
        constexpr int u = decltype(uu)::value;
        if (u < n) {
            unsigned long long key = 0;
            unsigned kbit = 0;
#pragma unroll
            for (int qq = 0; qq < QG; qq++) {
                const double xv = x[qq][u];
                const unsigned long long kk = (has[qq] && !((used >> qq) & 1u)) ? pivot_key(((warmbits >> qq) & 1u) ? xv * boost : xv, (unsigned long long)row[qq]) : 0ull;
                const bool better = kk > key;
                key = better ? kk : key;
                kbit = better ? (1u << qq) : kbit;
            }
            const unsigned long long wmax = wave_max_u64(key);
            double *slot = slots + (parity * (NT / 64) + (tid >> 6)) * NRS;
            if (key == wmax && key != 0) {
                static_for<0, QG>([&](auto qv) {
                    constexpr int qq = decltype(qv)::value;
                    if (kbit == (1u << qq)) {
#pragma unroll
                        for (int v = u; v < NRS; v++) { double t = x[qq][v]; asm volatile("" : "+v"(t)); slot[v] = t; }
                    }
                });
            }
            if ((tid & 63) == 0) red[parity * (NT / 64) + (tid >> 6)] = wmax;
            __syncthreads();
            unsigned long long best = red[parity * (NT / 64)];
            int wb = 0;
#pragma unroll
            for (int w2 = 1; w2 < NT / 64; w2++) {
                const unsigned long long r = red[parity * (NT / 64) + w2];
                if (r > best) { best = r; wb = w2; }
            }
            const double *win = slots + (parity * (NT / 64) + wb) * NRS;
            parity ^= 1;
            if (key == best && key != 0) { // the pivot row's owner: its multipliers are row u of L[rows]
                used |= kbit;
                static_for<0, QG>([&](auto qv) {
                    constexpr int qq = decltype(qv)::value;
                    if (kbit == (1u << qq)) {
                        mypos[qq] = u;
#pragma unroll
                        for (int j = 0; j < u; j++) { double t = x[qq][j]; asm volatile("" : "+v"(t)); Lr[u * LS + j] = t; }
                    }
                });
            }
            const double dp = win[u];
            if (tid == 0) { rows[u] = (int)(IDX_MASK - (best & IDX_MASK)); pivabs[u] = fabs(dp); }
            const double inv = dp != 0.0 ? 1.0 / dp : 0.0;
#pragma unroll
            for (int qq = 0; qq < QG; qq++)
                if (has[qq] && !((used >> qq) & 1u)) {
                    const double l = x[qq][u] * inv;
                    x[qq][u] = l;
#pragma unroll
                    for (int v = u + 1; v < NRS; v++) x[qq][v] -= l * win[v];
                }
        }
    });
    __syncthreads();
    CORE_STAMP(2);
    // ---- B = L inv(L[rows]): pivot rows are unit vectors, the others solve x Lr = l in registers
#pragma unroll
    for (int qq = 0; qq < QG; qq++) {
        if (has[qq] && ((used >> qq) & 1u)) {
#pragma unroll
            for (int t = 0; t < NRS; t++) x[qq][t] = (t == mypos[qq]) ? 1.0 : 0.0;
        } else if (has[qq]) {
            static_for<0, NRS>([&](auto jj) {
                constexpr int j = NRS - 1 - decltype(jj)::value;
                double sacc = x[qq][j];
#pragma unroll
                for (int t = j + 1; t < NRS; t++) sacc -= x[qq][t] * Lr[t * LS + j];
                x[qq][j] = sacc;
            });
        }
    }
    CORE_STAMP(3);
    // ---- maxvol: the swapped-in row goes round through LDS, every thread updates its registers
    int nswaps = 0;
    for (int it = 0; it < 200; it++) {
        unsigned long long key = 0;
#pragma unroll
        for (int qq = 0; qq < QG; qq++) {
            RowBest rb = {0ull, 0};
#pragma unroll
            for (int t = 0; t < NRS; t++) track(rb, x[qq][t], t);
            const unsigned long long kk = has[qq] ? row_key(rb, m, row[qq]) : 0ull;
            key = kk > key ? kk : key;
        }
        const unsigned long long best = block_max(key, red, parity);
        if (best == 0) break;
        const int lin = (int)(IDX_MASK - (best & IDX_MASK));
        const int bj = lin / m, bi = lin % m;
        double xbj[QG]; // the entries of column bj (a run-time column: selected, not indexed)
#pragma unroll
        for (int qq = 0; qq < QG; qq++) {
            xbj[qq] = x[qq][0];
#pragma unroll
            for (int t = 1; t < NRS; t++) xbj[qq] = (t == bj) ? x[qq][t] : xbj[qq];
        }
        static_for<0, QG>([&](auto qv) {
            constexpr int qq = decltype(qv)::value;
            if (row[qq] == bi) {
#pragma unroll
                for (int c = 0; c < NRS; c++) { double t = x[qq][c]; asm volatile("" : "+v"(t)); rowv[c] = t - (c == bj ? 1.0 : 0.0); }
                rowv[NRS] = xbj[qq];
            }
        });
        __syncthreads();
        const double piv = rowv[NRS];
        if (!(fabs(piv) > 1.0 + P.swap_tol)) break;
#pragma unroll
        for (int qq = 0; qq < QG; qq++)
            if (has[qq]) {
                const double cv = xbj[qq] / piv;
#pragma unroll
                for (int t = 0; t < NRS; t++) x[qq][t] -= cv * rowv[t];
            }
        if (tid == 0) rows[bj] = bi;
        nswaps++;
        // the barrier inside the next block_max stands between these reads of rowv and its next writer
    }
    __syncthreads();
    CORE_STAMP(4);
    sort_rows(n, rows, srows, pos);
    if (P.dir == 1) { // (a left-to-right step's core is never read: see core_step)
#pragma unroll
        for (int qq = 0; qq < QG; qq++)
            if (has[qq]) {
#pragma unroll
                for (int c = 0; c < NRS; c++) if (c < n) P.G[pos[c] + r0 * row[qq]] = x[qq][c]; // G[a' + r0 cc], cc = row
            }
    }
    CORE_STAMP(5);
    write_sets_and_next(P, n, srows, pivabs, nswaps);
    CORE_STAMP(6);
#ifdef C3SC_CORE_STAMPS
    if (threadIdx.x == 0 && P.stamps) P.stamps[7] += (unsigned long long)nswaps;
#endif
}

// The confirming iteration in one launch.  After an iteration that swapped rows, the next one usually changes nothing: with
// unchanged index sets all of its 2 d core steps are independent (each factors the fiber values its step already holds), so
// they run side by side, one workgroup per step, and only COMPARE the index sets they produce with the current ones.  If none
// differs, the right-to-left steps have written the iteration's cores and the sequential iteration -- same kernels, same
// inputs, same bits -- need not run; if one differs nothing was overwritten except the cores, which the sequential iteration
// that follows rewrites anyway.
template <bool INLDS>
__global__ void __launch_bounds__(NT) k_cross_confirm(const CoreArgs *steps) { core_step<INLDS>(steps[blockIdx.x]); }
template <int NR>
__global__ void __launch_bounds__(NT) k_cross_confirm_g(const CoreArgs *steps) { core_step_global<NR>(steps[blockIdx.x]); }

} // namespace

// =================================================================================================== host side
struct c3sc_cross_dev {
    int d = 0;
    int N[MAXD] = {0};
    int r[MAXD + 1] = {0};
    // one device slab: [sets I | sets J | idx | out | cores | work]
    char *slab = nullptr;
    size_t slab_bytes = 0;
    size_t offI[MAXD] = {0}, offJ[MAXD] = {0}, offG[MAXD] = {0}, off_idx = 0, off_out = 0, off_work = 0, sets_bytes = 0, cores_bytes = 0,
           off_cores = 0;
    // node memos: values of the current value-iteration sweep; policies (candidate indices) of the current policy iteration
    struct MemoTab {
        unsigned long long *keys = nullptr;
        double *vals = nullptr;
        size_t cap = 0;
        unsigned epoch = 0;
    } vmemo, pmemo;
    // per core step k: its fiber list and values stay on the device so that a step asked for the same list again in the same
    // generation is not recomputed (k_cross_idx / write_list_and_flag)
    size_t offIdx[MAXD] = {0}, offOut[MAXD] = {0}, off_flags = 0, off_steps = 0, work_stride = 0;
    unsigned long long gen = 0;
    long long policy_tag = -1; // the caller's policy-iteration counter the policy memo belongs to
    size_t offUidx[MAXD] = {0}; // per core step: [F_k][N_k] candidate indices between the policy pass and the evaluation pass (a block of
                                // its own per step: a step whose policy pass was skipped must find ITS policy there, not another step's)
    Strides strides;
    unsigned long long *counters = nullptr; // device [4]
    // pinned host staging for the one-copy upload / download
    char *stage = nullptr;
    size_t stage_bytes = 0;
    bool lds_optin = false;
    bool stage_fresh = false;                    // the pinned block holds the sets, cores and counters of the device as they are now
    unsigned long long pending[4] = {0, 0, 0, 0}; // counters read from the device but not handed to the caller yet
    int warm = 1;
    double swap_tol = 0.05;
    // streamed cores (c3sc_hip_cross_iteration_streamed): the right-to-left half sweep produces the cores in the order the host's
    // right-to-left orthogonalisation consumes them, so each is copied to the pinned block on a stream of its own as soon as its
    // step has run, and the host starts rounding while the later steps are still running
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_step[MAXD] = {nullptr}, ev_core[MAXD] = {nullptr};
    bool stream_cores = false;   // this iteration: copy each core early
    bool cores_streamed = false; // the pinned block already holds the cores of the last iteration
};

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

static int memo_shift_of(size_t cap)
{
    int shift = 64;
    for (size_t cp = cap; cp > 1; cp >>= 1) shift--;
    return shift;
}

// at least `want` slots (a power of two); keep = the entries of the current epoch move over (growth in the middle of a sweep)
static int memo_ensure(c3sc_hip_ctx *c, c3sc_cross_dev::MemoTab &t, size_t want, bool keep)
{
    if (want <= t.cap) return C3SC_OK;
    unsigned long long *nk = nullptr;
    double *nv = nullptr;
    HIPCHK(c, hipMalloc((void **)&nk, want * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc((void **)&nv, want * sizeof(double)));
    HIPCHK(c, hipMemset(nk, 0, want * sizeof(unsigned long long)));
    if (t.keys && keep && t.epoch != 0) {
        hipLaunchKernelGGL(k_cross_memo_rehash, dim3(256), dim3(256), 0, nullptr, t.keys, t.vals, (unsigned long long)t.cap, nk, nv,
                           (unsigned long long)(want - 1), memo_shift_of(want), (unsigned long long)t.epoch << 49);
        HIPCHK(c, hipDeviceSynchronize());
    } else {
        t.epoch = 0;
    }
    if (t.keys) HIPCHK(c, hipFree(t.keys));
    if (t.vals) HIPCHK(c, hipFree(t.vals));
    t.keys = nk; t.vals = nv; t.cap = want;
    return C3SC_OK;
}

static int memo_new_epoch(c3sc_hip_ctx *c, c3sc_cross_dev::MemoTab &t)
{
    t.epoch++;
    if (t.epoch > 0x7FFF) { // the epoch field wrapped: really clear
        HIPCHK(c, hipMemset(t.keys, 0, t.cap * sizeof(unsigned long long)));
        t.epoch = 1;
    }
    return C3SC_OK;
}

extern "C" {

void c3sc_hip_cross_free(c3sc_hip_ctx *c)
{
    if (!c || !c->cross) return;
    c3sc_cross_dev *x = c->cross;
    if (x->slab) (void)hipFree(x->slab);
    for (auto *t : {&x->vmemo, &x->pmemo}) {
        if (t->keys) (void)hipFree(t->keys);
        if (t->vals) (void)hipFree(t->vals);
    }
    if (x->counters) (void)hipFree(x->counters);
    if (x->stage) (void)hipHostFree(x->stage);
    for (int k = 0; k < MAXD; k++) {
        if (x->ev_step[k]) (void)hipEventDestroy(x->ev_step[k]);
        if (x->ev_core[k]) (void)hipEventDestroy(x->ev_core[k]);
    }
    if (x->copy_stream) (void)hipStreamDestroy(x->copy_stream);
    delete x;
    c->cross = nullptr;
}

int c3sc_hip_cross_setup(c3sc_hip_ctx *c, const size_t *ranks, const int32_t *const *I, const int32_t *const *J, int new_sweep)
{
    if (!c || c->d == 0 || !ranks || !I || !J) return fail(c, C3SC_ERR_ARG, "cross_setup: set_grid first");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->cross) c->cross = new c3sc_cross_dev();
    c3sc_cross_dev *x = c->cross;
    const int d = c->d;
    if (ranks[0] != 1 || ranks[d] != 1) return fail(c, C3SC_ERR_ARG, "cross_setup: ranks[0] and ranks[d] must be 1");
    size_t fmax = 1, nmax = 1, wmax = 1;
    x->d = d;
    for (int k = 0; k <= d; k++) {
        if (ranks[k] < 1 || ranks[k] > MAXR) return fail(c, C3SC_ERR_UNSUPPORTED, "cross_setup: ranks up to 48");
        x->r[k] = (int)ranks[k];
    }
    // layout of the slab for these ranks
    size_t off = 0;
    for (int k = 0; k < d; k++) {
        x->N[k] = c->ngrid[k];
        x->offI[k] = off; off += up256((size_t)x->r[k] * (k ? k : 1) * sizeof(int32_t));
    }
    for (int k = 0; k < d; k++) { x->offJ[k] = off; off += up256((size_t)x->r[k + 1] * ((d - 1 - k) ? (d - 1 - k) : 1) * sizeof(int32_t)); }
    x->sets_bytes = off;
    x->off_cores = off;
    for (int k = 0; k < d; k++) {
        const size_t F = (size_t)x->r[k] * x->r[k + 1], sz = F * x->N[k];
        fmax = std::max(fmax, F); nmax = std::max(nmax, (size_t)x->N[k]); wmax = std::max(wmax, sz);
        if (sz > (IDX_MASK + 1) / 2 || (size_t)x->r[k] * x->N[k] > MAXROWS || (size_t)x->N[k] * x->r[k + 1] > MAXROWS)
            return fail(c, C3SC_ERR_UNSUPPORTED, "cross_setup: core too large for the one-workgroup factorisation");
        x->offG[k] = off; off += up256(sz * sizeof(double));
    }
    x->cores_bytes = off - x->off_cores;
    x->off_idx = off; off += up256(fmax * d * sizeof(int32_t));
    x->off_out = off; off += up256((fmax + 64) * nmax * sizeof(double)); // + the padding rows of a sharded step (world * ceil(F / world) >= F)
    for (int k = 0; k < d; k++) {
        const size_t F = (size_t)x->r[k] * x->r[k + 1];
        x->offIdx[k] = off; off += up256(F * d * sizeof(int32_t));
        x->offOut[k] = off; off += up256((F + 64) * x->N[k] * sizeof(double));
    }
    x->off_flags = off; off += up256(MAXD * (sizeof(int) + sizeof(unsigned long long)) + 64);
    x->off_steps = off; off += up256(2 * MAXD * sizeof(CoreArgs) + 64);
    // scratch of a factorisation that does not fit LDS: one block per core step of an iteration (the batched confirmation runs all
    // 2 d steps side by side; the sequential iteration uses the first block)
    for (int k = 0; k < d; k++) { // the global-scratch core step pads the columns of its matrix to 32 / 40 / 48 (core_step_global)
        const size_t rmaxk = (size_t)std::max(x->r[k], x->r[k + 1]), rowsk = rmaxk * x->N[k];
        wmax = std::max(wmax, rowsk * (rmaxk <= 32 ? 32 : rmaxk <= 40 ? 40 : 48));
    }
    x->work_stride = up256(wmax * sizeof(double));
    x->off_work = off; off += 2 * (size_t)d * x->work_stride;
    for (int k = 0; k < d; k++) { x->offUidx[k] = off; off += up256(((size_t)x->r[k] * x->r[k + 1] + 64) * x->N[k] * sizeof(int32_t)); }
    if (off > x->slab_bytes) {
        if (x->slab) HIPCHK(c, hipFree(x->slab));
        x->slab = nullptr; x->slab_bytes = 0;
        HIPCHK(c, hipMalloc((void **)&x->slab, off));
        x->slab_bytes = off;
    }
    x->stage_fresh = false;
    x->gen++; // every set-up starts a new generation of cached step values: the layout, the ranks or the sweep changed
    HIPCHK(c, hipMemsetAsync(x->slab + x->off_flags, 0, MAXD * (sizeof(int) + sizeof(unsigned long long)) + 64, nullptr));
    const size_t need_stage = x->sets_bytes + x->cores_bytes + 64 + 2 * MAXD * sizeof(CoreArgs) + 64;
    if (need_stage > x->stage_bytes) {
        if (x->stage) HIPCHK(c, hipHostFree(x->stage));
        x->stage = nullptr; x->stage_bytes = 0;
        HIPCHK(c, hipHostMalloc((void **)&x->stage, need_stage, hipHostMallocDefault));
        x->stage_bytes = need_stage;
    }
    if (!x->counters) {
        HIPCHK(c, hipMalloc((void **)&x->counters, 4 * sizeof(unsigned long long)));
        HIPCHK(c, hipMemset(x->counters, 0, 4 * sizeof(unsigned long long)));
    }
    // value memo: every node a sweep can touch (5 iterations x 2 half sweeps x sum_k F_k N_k, before rank kicks) at load <= 1/3
    size_t nodes = 0;
    for (int k = 0; k < d; k++) nodes += (size_t)x->r[k] * x->r[k + 1] * x->N[k];
    // C3SC_MEMO_MIN_LOG2 / C3SC_MEMO_SCALE: test hooks that make the table small enough to fill (tests/test_solver_loops.py)
    const size_t memo_scale = getenv("C3SC_MEMO_SCALE") ? (size_t)atoi(getenv("C3SC_MEMO_SCALE")) : 32;
    size_t want = (size_t)1 << (getenv("C3SC_MEMO_MIN_LOG2") ? atoi(getenv("C3SC_MEMO_MIN_LOG2")) : 16);
    while (want < memo_scale * nodes) want <<= 1;
    if (want < x->vmemo.cap) want = x->vmemo.cap; // never below what a growth (c3sc_hip_cross_grow_memo) has reached
    {
        const bool fresh = new_sweep || x->vmemo.epoch == 0;
        int rc = memo_ensure(c, x->vmemo, want, !fresh);
        if (rc != C3SC_OK) return rc;
        if (fresh || x->vmemo.epoch == 0) { rc = memo_new_epoch(c, x->vmemo); if (rc != C3SC_OK) return rc; }
    }
    long long s = 1;
    for (int m = d - 1; m >= 0; m--) { x->strides.s[m] = s; s *= x->N[m]; }
    if ((unsigned long long)s > ID_MASK) return fail(c, C3SC_ERR_UNSUPPORTED, "cross_setup: more than 2^48 grid nodes");
    // index sets: one upload
    std::memset(x->stage, 0, x->sets_bytes);
    for (int k = 0; k < d; k++) {
        std::memcpy(x->stage + x->offI[k], I[k], (size_t)x->r[k] * k * sizeof(int32_t));
        std::memcpy(x->stage + x->offJ[k], J[k], (size_t)x->r[k + 1] * (d - 1 - k) * sizeof(int32_t));
    }
    HIPCHK(c, hipMemcpy(x->slab, x->stage, x->sets_bytes, hipMemcpyHostToDevice));
    if (!x->lds_optin) {
        HIPCHK(c, hipFuncSetAttribute((const void *)k_cross_core<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP_BYTES));
        HIPCHK(c, hipFuncSetAttribute((const void *)k_cross_confirm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP_BYTES));
        x->lds_optin = true;
    }
    // The set-up ran on the NULL stream (flag / tag reset, index-set upload, memo growth); the iteration, confirmation and fetch
    // calls take the CALLER's stream, which need not be ordered against it (a non-blocking stream is not).  Everything issued here
    // is complete before the call returns, so any stream may follow.
    HIPCHK(c, hipStreamSynchronize(nullptr));
    return C3SC_OK;
}

/* The node memo (and the policy memo, if one exists) reported an overflow (info[3] == 1 of c3sc_hip_cross_fetch): double the tables,
 * keeping the entries of the current epoch, so that the following iterations of this sweep find room.  Values computed while the
 * table was full are correct fiber values that merely were not stored. */
int c3sc_hip_cross_grow_memo(c3sc_hip_ctx *c)
{
    if (!c || !c->cross) return fail(c, C3SC_ERR_ARG, "cross_grow_memo: cross_setup first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    c3sc_cross_dev *x = c->cross;
    int rc = C3SC_OK;
    if (x->vmemo.cap) rc = memo_ensure(c, x->vmemo, 2 * x->vmemo.cap, true);
    if (rc == C3SC_OK && x->pmemo.cap) rc = memo_ensure(c, x->pmemo, 2 * x->pmemo.cap, true);
    return rc;
}

/* pivot-search options of the core steps (defaults: warm start on, swap tolerance 0.05 -- the host driver's) */
int c3sc_hip_cross_options(c3sc_hip_ctx *c, int warm_pivots, double swap_tol)
{
    if (!c || !(swap_tol >= 0.0)) return fail(c, C3SC_ERR_ARG, "cross_options: bad arguments");
    if (!c->cross) c->cross = new c3sc_cross_dev();
    c->cross->warm = warm_pivots ? 1 : 0;
    c->cross->swap_tol = swap_tol;
    return C3SC_OK;
}

/* one cross iteration: left-to-right half sweep (new left index sets), then right-to-left (new right index sets); the cores
 * of the right-to-left half sweep are the iteration's result.  pol == null: the fibers are bellman_vi's (value memo);
 * pol != null: bellman_pi's -- per core step the greedy policy of pol's value function (policy memo, first entry stays for
 * the whole policy iteration `policy_tag`), then the evaluation of that policy on c's value function (no value memo: the
 * reference's is never hit). */
// the policy memo of policy iteration `policy_tag` (first entry stays for the whole policy iteration)
static int prepare_policy_memo(c3sc_hip_ctx *c, c3sc_hip_ctx *pol, long long policy_tag, int box)
{
    c3sc_cross_dev *x = c->cross;
    const int d = x->d;
    if (box) return fail(c, C3SC_ERR_UNSUPPORTED, "cross_iteration_pi: candidate lists only");
    if (pol->d != d) return fail(c, C3SC_ERR_ARG, "cross_iteration_pi: the policy context has another grid");
    size_t nodes = 0;
    for (int k = 0; k < d; k++) nodes += (size_t)x->r[k] * x->r[k + 1] * x->N[k];
    size_t want = 1 << 16;
    while (want < 256 * nodes) want <<= 1; // a policy iteration runs ~10 evaluation sweeps over one policy memo
    const bool fresh = policy_tag != x->policy_tag || x->pmemo.epoch == 0;
    int rc = memo_ensure(c, x->pmemo, want, !fresh);
    if (rc != C3SC_OK) return rc;
    if (fresh) { rc = memo_new_epoch(c, x->pmemo); if (rc != C3SC_OK) return rc; x->policy_tag = policy_tag; }
    return C3SC_OK;
}

// The fibers of core step k from its current list (x->offIdx[k]) into the step's block (x->offOut[k]), memo applied: bellman_vi's
// (pol == null, value memo) or bellman_pi's (greedy policy of pol's value function through the policy memo, then its evaluation
// on c's value function).  Sharded contexts evaluate their block of rows and all-gather in place.
static int step_fibers(c3sc_hip_ctx *c, c3sc_hip_ctx *pol, int box, int k, void *stream)
{
    c3sc_cross_dev *x = c->cross;
    const int d = x->d;
    hipStream_t st = (hipStream_t)stream;
    c3sc_hip_comm *comm = c->shard_comm;
    const int world = comm ? c3sc_hip_comm_world(comm) : 1, rank = comm ? c3sc_hip_comm_rank(comm) : 0;
    c3sc_cross_dev::MemoTab &mt = pol ? x->pmemo : x->vmemo;
    const int shift = memo_shift_of(mt.cap);
    int32_t *uidx = (int32_t *)(x->slab + x->offUidx[k]);
    int *skipf = (int *)(x->slab + x->off_flags);
    const int r0 = x->r[k], r1 = x->r[k + 1], N = x->N[k];
    const size_t F = (size_t)r0 * r1;
    const long total = (long)F * N;
    int32_t *idx = (int32_t *)(x->slab + x->offIdx[k]);
    double *out = (double *)(x->slab + x->offOut[k]);
    // a step that already holds the values of this very list (flag written by the kernel that made the list) is not recomputed
    c->skip_flag = comm ? nullptr : skipf + k;
    if (pol) pol->skip_flag = c->skip_flag;
    // the fiber-per-wave kernel applies the memo in its epilogue (ctx->memo); any other kernel is followed by the memo pass
    c3sc_hip_ctx *mc = pol ? pol : c; // the context whose launch carries the memo
    mc->memo.keys = mt.keys; mc->memo.vals = mt.vals; mc->memo.capmask = (unsigned long long)(mt.cap - 1); mc->memo.shift = shift;
    mc->memo.epoch_bits = (unsigned long long)mt.epoch << 49; mc->memo.counters = x->counters; mc->memo.mode = pol ? 1 : 0;
    for (int m = 0; m < d; m++) mc->memo.stride[m] = x->strides.s[m];
    mc->memo.applied = false;
    int rc;
    if (comm) { // sharded: this rank's block of rows, then the all-gather in place; the memo pass runs on the full array,
                // in the same order on every rank, so that all ranks keep identical memos and take identical decisions
        const size_t per = (F + world - 1) / world, lo = std::min(F, (size_t)rank * per), hi = std::min(F, lo + per);
        mc->memo.keys = nullptr;
        rc = C3SC_OK;
        if (hi > lo)
            rc = box ? c3sc_hip_bellman_fibers_box(c, k, hi - lo, idx + lo * d, out + lo * N, nullptr, nullptr, stream)
                     : c3sc_hip_bellman_fibers(c, k, hi - lo, idx + lo * d, out + lo * N, nullptr, nullptr, stream);
        if (getenv("C3SC_INJECT_SHARD_FAILURE")) rc = fail(c, C3SC_ERR_HIP, "cross_iteration: injected failure of this rank's fibers");
        // A rank whose launch failed still ENTERS the all-gather -- its peers are about to wait in it -- with its rows set to NaN
        // (0xFF bytes; no fiber value ever is one).  The memo pass that follows on every rank sees the mark and raises
        // counters[3] = 2, so that all ranks fail this sweep together instead of hanging (c3sc_cross.c: sharded_fibers_idx does the
        // same on the host-driven path).
        const int local_rc = rc;
        if (local_rc != C3SC_OK) (void)hipMemsetAsync(out + (size_t)rank * per * N, 0xFF, per * N * sizeof(double), st);
        rc = c3sc_hip_comm_allgather(comm, out + (size_t)rank * per * N, out, per * N, stream);
        if (local_rc != C3SC_OK) {
            if (rc == C3SC_OK)
                hipLaunchKernelGGL(k_cross_memo, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, idx, out, total, N, d, k, x->strides, mt.keys,
                                   mt.vals, (unsigned long long)(mt.cap - 1), shift, (unsigned long long)mt.epoch << 49, x->counters);
            rc = local_rc;
        }
    } else if (pol) rc = c3sc_hip_bellman_fibers(pol, k, F, idx, out, uidx, nullptr, stream); // values unused: the policy is the output
    else rc = box ? c3sc_hip_bellman_fibers_box(c, k, F, idx, out, nullptr, nullptr, stream)
                  : c3sc_hip_bellman_fibers(c, k, F, idx, out, nullptr, nullptr, stream);
    mc->memo.keys = nullptr;
    if (pol) pol->skip_flag = nullptr;
    if (rc != C3SC_OK) { c->skip_flag = nullptr; if (pol) c->err = pol->err; return rc; }
    if (pol) {
        if (!mc->memo.applied) { c->skip_flag = nullptr; return fail(c, C3SC_ERR_UNSUPPORTED, "cross_iteration_pi: the policy pass needs the fiber-per-wave kernel"); }
        rc = c3sc_hip_policy_fibers(c, k, F, idx, uidx, out, nullptr, stream);
        c->skip_flag = nullptr;
        if (rc != C3SC_OK) return rc;
    } else if (!mc->memo.applied)
        hipLaunchKernelGGL(k_cross_memo, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, idx, out, total, N, d, k, x->strides, mt.keys,
                           mt.vals, (unsigned long long)(mt.cap - 1), shift, (unsigned long long)mt.epoch << 49, x->counters);
    c->skip_flag = nullptr;
    return C3SC_OK;
}

static NextList next_list(c3sc_cross_dev *x, int k, bool enable)
{
    int *skipf = (int *)(x->slab + x->off_flags);
    unsigned long long *tags = (unsigned long long *)(x->slab + x->off_flags + 64);
    NextList L;
    L.idx = (int32_t *)(x->slab + x->offIdx[k]); L.I = (int32_t *)(x->slab + x->offI[k]); L.J = (int32_t *)(x->slab + x->offJ[k]);
    L.r0 = x->r[k]; L.r1 = x->r[k + 1]; L.k = k; L.d = x->d;
    L.skip = skipf + k; L.tag = tags + k; L.gen = x->gen; L.enable = enable ? 1 : 0;
    return L;
}

static int cross_iteration_impl(c3sc_hip_ctx *c, c3sc_hip_ctx *pol, long long policy_tag, int box, void *stream)
{
    if (!c || !c->cross || c->cross->d == 0) return fail(c, C3SC_ERR_ARG, "cross_iteration: cross_setup first");
    c3sc_cross_dev *x = c->cross;
    x->stage_fresh = false;
    if (x->cores_streamed) { // a streamed iteration whose result was never fetched: its copies must not land in the middle of this one's
        HIPCHK(c, hipStreamSynchronize(x->copy_stream));
        x->cores_streamed = false;
    }
    const int d = x->d;
    hipStream_t st = (hipStream_t)stream;
    c3sc_hip_comm *comm = c->shard_comm;
    const int world = comm ? c3sc_hip_comm_world(comm) : 1;
    if (comm && pol) return fail(c, C3SC_ERR_UNSUPPORTED, "cross_iteration_pi: not sharded (use the host-driven driver)");
    if (world > 64) return fail(c, C3SC_ERR_UNSUPPORTED, "cross_iteration: up to 64 ranks");
    if (pol) {
        const int rc = prepare_policy_memo(c, pol, policy_tag, box);
        if (rc != C3SC_OK) return rc;
    }
    auto setI = [&](int k) { return (int32_t *)(x->slab + x->offI[k]); };
    auto setJ = [&](int k) { return (int32_t *)(x->slab + x->offJ[k]); };
    auto outOf = [&](int k) { return (double *)(x->slab + x->offOut[k]); };
    auto nextList = [&](int k) { return next_list(x, k, comm == nullptr); };
    // the first step's fiber list; every later one is written by the core step before it
    hipLaunchKernelGGL(k_cross_idx, dim3(1), dim3(256), 0, st, nextList(0));
    for (int s = 0; s < 2 * d; s++) {
        const int half = s / d, k = half == 0 ? s : 2 * d - 1 - s;
        const int r0 = x->r[k], r1 = x->r[k + 1], N = x->N[k];
        const size_t F = (size_t)r0 * r1;
        double *out = outOf(k);
        const int rc = step_fibers(c, pol, box, k, stream);
        if (rc != C3SC_OK) return rc;
        CoreArgs P;
        P.out = out; P.r0 = r0; P.N = N; P.r1 = r1; P.k = k; P.d = d;
        P.dir = half;
        P.copy_only = (half == 0) ? (k == d - 1) : (k == 0);
        P.set_in = half == 0 ? setI(k) : setJ(k);
        P.set_out = P.copy_only ? nullptr : (half == 0 ? setI(k + 1) : setJ(k - 1));
        P.G = (double *)(x->slab + x->offG[k]);
        P.work = (double *)(x->slab + x->off_work);
        P.counters = x->counters;
        P.warm = x->warm;
        P.swap_tol = x->swap_tol;
        P.confirm = 0; P.mismatch = nullptr;
#ifdef C3SC_CORE_STAMPS
        static unsigned long long *g_stamps = nullptr;
        if (!g_stamps) { (void)hipMalloc((void **)&g_stamps, 64 * 16 * sizeof(unsigned long long)); (void)hipMemset(g_stamps, 0, 64 * 16 * sizeof(unsigned long long)); }
        P.stamps = g_stamps + 16 * s;
        if (s == 2 * d - 1) {
            static int calls = 0;
            if (++calls % 20 == 0) {
                (void)hipStreamSynchronize(st);
                unsigned long long hs[64 * 16];
                (void)hipMemcpy(hs, g_stamps, sizeof(hs), hipMemcpyDeviceToHost);
                for (int q = 0; q < 2 * d; q++) // the mean over the last 20 iterations (the last step of the 20th is still running: 19 of it)
                    fprintf(stderr, "core step %2d (iterations %d-%d): load %6.0f warm %6.0f lu %6.0f backsub %6.0f maxvol %6.0f results %6.0f sets+next %6.0f cycles, swaps %.2f\n", q, calls - 19, calls,
                            hs[16 * q + 0] / 20.0, hs[16 * q + 1] / 20.0, hs[16 * q + 2] / 20.0, hs[16 * q + 3] / 20.0,
                            hs[16 * q + 4] / 20.0, hs[16 * q + 5] / 20.0, hs[16 * q + 6] / 20.0, hs[16 * q + 7] / 20.0);
                for (int q = 0; q < 2 * d; q++) // global-scratch steps: the LU's columns as thread 0 sees them
                    if (hs[16 * q + 8]) fprintf(stderr, "core step %2d LU columns: own rows %6.0f pivot search %6.0f pivot row %6.0f scale + store %6.0f cycles\n", q,
                                                hs[16 * q + 8] / 20.0, hs[16 * q + 9] / 20.0, hs[16 * q + 10] / 20.0, hs[16 * q + 11] / 20.0);
                (void)hipMemset(g_stamps, 0, 64 * 16 * sizeof(unsigned long long));
            }
        }
#endif
        std::memset(&P.next, 0, sizeof(P.next));
        if (s + 1 < 2 * d) {
            const int nh = (s + 1) / d, nk = nh == 0 ? s + 1 : 2 * d - 2 - s;
            P.next = nextList(nk);
        }
        const size_t mn = F * N * sizeof(double);
        if (P.copy_only) hipLaunchKernelGGL(k_cross_core<true>, dim3(1), dim3(NT), 0, st, P);
        else if (mn <= LDS_CAP_BYTES && r0 <= LDSR && r1 <= LDSR) hipLaunchKernelGGL(k_cross_core<true>, dim3(1), dim3(NT), mn, st, P);
        else { // global scratch; the rows' register image is padded to the next of 32 / 40 / 48 columns
            const int ncol = P.dir == 0 ? r1 : r0;
            const size_t lds = (size_t)MAXR * MAXR * sizeof(double);
            if (ncol <= 32) hipLaunchKernelGGL(k_cross_core_g<32>, dim3(1), dim3(NT), lds, st, P);
            else if (ncol <= 40) hipLaunchKernelGGL(k_cross_core_g<40>, dim3(1), dim3(NT), lds, st, P);
            else hipLaunchKernelGGL(k_cross_core_g<48>, dim3(1), dim3(NT), lds, st, P);
        }
        if (x->stream_cores && half == 1) { // the core of this step is final: on its way to the host behind the step, beside the next steps
            HIPCHK(c, hipEventRecord(x->ev_step[k], st));
            HIPCHK(c, hipStreamWaitEvent(x->copy_stream, x->ev_step[k], 0));
            HIPCHK(c, hipMemcpyAsync(x->stage + x->offG[k], x->slab + x->offG[k], (size_t)r0 * N * r1 * sizeof(double), hipMemcpyDeviceToHost, x->copy_stream));
            HIPCHK(c, hipEventRecord(x->ev_core[k], x->copy_stream));
        }
    }
    HIPCHK(c, hipGetLastError());
    if (x->stream_cores) { x->cores_streamed = true; x->stream_cores = false; }
    return C3SC_OK;
}

int c3sc_hip_cross_iteration(c3sc_hip_ctx *c, int box, void *stream) { return cross_iteration_impl(c, nullptr, 0, box, stream); }

int c3sc_hip_cross_iteration_streamed(c3sc_hip_ctx *c, int box, void *stream)
{
    if (!c || !c->cross || c->cross->d == 0) return fail(c, C3SC_ERR_ARG, "cross_iteration_streamed: cross_setup first");
    c3sc_cross_dev *x = c->cross;
    HIPCHK(c, hipSetDevice(c->device));
    if (!x->copy_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&x->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < MAXD; k++) {
            HIPCHK(c, hipEventCreateWithFlags(&x->ev_step[k], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&x->ev_core[k], hipEventDisableTiming));
        }
    }
    x->stream_cores = true;
    const int rc = cross_iteration_impl(c, nullptr, 0, box, stream);
    x->stream_cores = false;
    return rc;
}

int c3sc_hip_cross_wait_core(c3sc_hip_ctx *c, int k, double *h_core)
{
    if (!c || !c->cross || k < 0 || k >= c->cross->d || !h_core) return fail(c, C3SC_ERR_ARG, "cross_wait_core: bad argument");
    c3sc_cross_dev *x = c->cross;
    if (!x->cores_streamed) return fail(c, C3SC_ERR_ARG, "cross_wait_core: no streamed iteration is under way");
    HIPCHK(c, hipEventSynchronize(x->ev_core[k]));
    std::memcpy(h_core, x->stage + x->offG[k], (size_t)x->r[k] * x->N[k] * x->r[k + 1] * sizeof(double));
    return C3SC_OK;
}

int c3sc_hip_cross_iteration_pi(c3sc_hip_ctx *c, c3sc_hip_ctx *policy_ctx, long long policy_tag, void *stream)
{
    if (!policy_ctx) return fail(c, C3SC_ERR_ARG, "cross_iteration_pi: null policy context");
    return cross_iteration_impl(c, policy_ctx, policy_tag, 0, stream);
}

/* The confirming iteration in one launch (k_cross_confirm): valid after a complete c3sc_hip_cross_iteration[_pi] -- every
 * step then holds the fiber values of its current index sets.  *confirmed = 1: all 2 d steps reproduced their index sets; the
 * cores of the iteration are in place (fetch them), nothing else changed.  *confirmed = 0: run the ordinary iteration. */
static int stage_download(c3sc_hip_ctx *c, const void *extra, void *extra_host, size_t extra_bytes, hipStream_t st);

int c3sc_hip_cross_confirm(c3sc_hip_ctx *c, int *confirmed, void *stream)
{
    if (!c || !c->cross || c->cross->d == 0 || !confirmed) return fail(c, C3SC_ERR_ARG, "cross_confirm: cross_setup first");
    c3sc_cross_dev *x = c->cross;
    const int d = x->d;
    hipStream_t st = (hipStream_t)stream;
    *confirmed = 0;
    CoreArgs *h = (CoreArgs *)(x->stage + x->sets_bytes + x->cores_bytes + 64);
    int *mismatch = (int *)(x->slab + x->off_flags) + MAXD + 1;
    size_t maxmn = 0;
    for (int s = 0; s < 2 * d; s++) {
        const int half = s / d, k = half == 0 ? s : 2 * d - 1 - s;
        CoreArgs &P = h[s];
        std::memset(&P, 0, sizeof(P));
        P.out = (double *)(x->slab + x->offOut[k]);
        P.r0 = x->r[k]; P.N = x->N[k]; P.r1 = x->r[k + 1]; P.k = k; P.d = d;
        P.dir = half;
        P.copy_only = (half == 0) ? (k == d - 1) : (k == 0);
        P.set_in = (int32_t *)(x->slab + (half == 0 ? x->offI[k] : x->offJ[k]));
        P.set_out = P.copy_only ? nullptr : (int32_t *)(x->slab + (half == 0 ? x->offI[k + 1] : x->offJ[k - 1]));
        P.G = (double *)(x->slab + x->offG[k]);
        P.work = (double *)(x->slab + x->off_work + (size_t)s * x->work_stride); // used when a step does not fit LDS
        P.counters = x->counters;
        P.warm = x->warm;
        P.swap_tol = x->swap_tol;
        P.confirm = 1;
        P.mismatch = mismatch;
#ifdef C3SC_CORE_STAMPS
        P.stamps = nullptr;
#endif
        if (!P.copy_only) maxmn = std::max(maxmn, (size_t)P.r0 * P.r1 * P.N * sizeof(double));
        if (P.r0 > LDSR || P.r1 > LDSR) maxmn = LDS_CAP_BYTES + 1; // a rank above 32: the global-scratch form for all steps
    }
    HIPCHK(c, hipMemsetAsync(mismatch, 0, sizeof(int), st));
    HIPCHK(c, hipMemcpyAsync(x->slab + x->off_steps, h, 2 * d * sizeof(CoreArgs), hipMemcpyHostToDevice, st));
    // every step in LDS, or -- if one of them does not fit (rank 16 on 101 nodes is 207 KB) -- every step on its own block of
    // global scratch: the same code either way (core_step<INLDS>), as in the sequential iteration
    if (maxmn <= LDS_CAP_BYTES)
        hipLaunchKernelGGL(k_cross_confirm<true>, dim3(2 * d), dim3(NT), maxmn, st, (const CoreArgs *)(x->slab + x->off_steps));
    else {
        int maxr = 1;
        for (int k = 0; k <= d; k++) maxr = std::max(maxr, x->r[k]);
        const size_t lds = (size_t)MAXR * MAXR * sizeof(double);
        const CoreArgs *steps = (const CoreArgs *)(x->slab + x->off_steps);
        if (maxr <= 32) hipLaunchKernelGGL(k_cross_confirm_g<32>, dim3(2 * d), dim3(NT), lds, st, steps);
        else if (maxr <= 40) hipLaunchKernelGGL(k_cross_confirm_g<40>, dim3(2 * d), dim3(NT), lds, st, steps);
        else hipLaunchKernelGGL(k_cross_confirm_g<48>, dim3(2 * d), dim3(NT), lds, st, steps);
    }
    HIPCHK(c, hipGetLastError());
    // the flag and, in the same wait, everything a successful confirmation is followed by (c3sc_hip_cross_fetch then copies from
    // the pinned block); after a mismatch the block is stale and the counters it brought wait in `pending`
    int *hflag = (int *)(x->stage + x->sets_bytes + x->cores_bytes + 4 * sizeof(unsigned long long) + sizeof(unsigned) + 4);
    *hflag = 1;
    const int rc = stage_download(c, mismatch, hflag, sizeof(int), st);
    if (rc != C3SC_OK) return rc;
    *confirmed = *hflag == 0;
    if (!*confirmed) x->stage_fresh = false;
    return C3SC_OK;
}

/* A whole iteration in d + 1 launches, for a sweep that starts from index sets believed to be its fixed point already (the
 * previous sweep's: the value function moved a little, the warm-started pivot search usually picks the same rows).  The fiber
 * lists of all d cores are formed from the CURRENT sets and evaluated back to back -- no factorisation in between, so nothing
 * waits for a core step -- and the confirming launch (c3sc_hip_cross_confirm) factors all 2 d steps side by side.  *confirmed = 1:
 * every step reproduced its index set, i.e. the sequential iteration would have asked for exactly these fibers in exactly this
 * order (left-to-right lists, the right-to-left half all memo hits) and returned these cores: same kernels, same inputs, same
 * bits.  *confirmed = 0: run c3sc_hip_cross_iteration; the lists evaluated here stay cached (same generation), so only the steps
 * whose sets really change are evaluated again.  Value iteration, unsharded; otherwise *confirmed = 0 and nothing is launched. */
int c3sc_hip_cross_speculate(c3sc_hip_ctx *c, c3sc_hip_ctx *pol, long long policy_tag, int box, int *confirmed, void *stream)
{
    if (!c || !c->cross || c->cross->d == 0 || !confirmed) return fail(c, C3SC_ERR_ARG, "cross_speculate: cross_setup first");
    *confirmed = 0;
    c3sc_cross_dev *x = c->cross;
    x->stage_fresh = false;
    if (c->shard_comm) return C3SC_OK;
    const int d = x->d;
    hipStream_t st = (hipStream_t)stream;
    if (pol) {
        const int rc = prepare_policy_memo(c, pol, policy_tag, box);
        if (rc != C3SC_OK) return rc;
    }
    for (int k = 0; k < d; k++) {
        hipLaunchKernelGGL(k_cross_idx, dim3(1), dim3(256), 0, st, next_list(x, k, true));
        const int rc = step_fibers(c, pol, box, k, stream);
        if (rc != C3SC_OK) return rc;
    }
    HIPCHK(c, hipGetLastError());
    return c3sc_hip_cross_confirm(c, confirmed, stream);
}

/* wait for the iteration and bring back: cores (working layout G[a + r_k (j + N_k b)]), both families of index sets, and
 * info = {nodes stored in the memo since the last fetch, rank-deficient factorisation seen, maxvol swaps, memo overflow} */
// device -> pinned block: index sets, cores, the four counters (then cleared on the device) and the context's status word; one
// wait.  `extra` / `extra_bytes`: one more small device word to bring along (the confirming launch's mismatch flag).
static int stage_download(c3sc_hip_ctx *c, const void *extra, void *extra_host, size_t extra_bytes, hipStream_t st)
{
    c3sc_cross_dev *x = c->cross;
    const size_t nb = x->sets_bytes + x->cores_bytes;
    unsigned *hstat = (unsigned *)(x->stage + nb + 4 * sizeof(unsigned long long));
    if (x->cores_streamed) { // the cores came over one by one behind their steps (and the host may be reading them): the sets only
        HIPCHK(c, hipStreamSynchronize(x->copy_stream));
        HIPCHK(c, hipMemcpyAsync(x->stage, x->slab, x->off_cores, hipMemcpyDeviceToHost, st));
        x->cores_streamed = false;
    } else
        HIPCHK(c, hipMemcpyAsync(x->stage, x->slab, nb, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(x->stage + nb, x->counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemsetAsync(x->counters, 0, 4 * sizeof(unsigned long long), st));
    HIPCHK(c, hipMemcpyAsync(hstat, c->d_status, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    if (extra) HIPCHK(c, hipMemcpyAsync(extra_host, extra, extra_bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const unsigned long long *cnt = (const unsigned long long *)(x->stage + nb);
    for (int i = 0; i < 4; i++) x->pending[i] += cnt[i];
    c->status_cache = *hstat;
    c->status_cache_valid = true; // until the next launch through this context
    x->stage_fresh = true;
    return C3SC_OK;
}

int c3sc_hip_cross_fetch(c3sc_hip_ctx *c, double *const *h_cores, int32_t *const *h_I, int32_t *const *h_J, unsigned long long *info,
                         void *stream)
{
    if (!c || !c->cross) return fail(c, C3SC_ERR_ARG, "cross_fetch: cross_setup first");
    c3sc_cross_dev *x = c->cross;
    if (!x->stage_fresh) { // a confirming launch that succeeded has brought everything over already
        const int rc = stage_download(c, nullptr, nullptr, 0, (hipStream_t)stream);
        if (rc != C3SC_OK) return rc;
    }
    x->stage_fresh = false;
    const int d = x->d;
    for (int k = 0; k < d; k++) {
        if (h_cores) std::memcpy(h_cores[k], x->stage + x->offG[k], (size_t)x->r[k] * x->N[k] * x->r[k + 1] * sizeof(double));
        if (h_I) std::memcpy(h_I[k], x->stage + x->offI[k], (size_t)x->r[k] * k * sizeof(int32_t));
        if (h_J) std::memcpy(h_J[k], x->stage + x->offJ[k], (size_t)x->r[k + 1] * (d - 1 - k) * sizeof(int32_t));
    }
    if (info) std::memcpy(info, x->pending, 4 * sizeof(unsigned long long));
    std::memset(x->pending, 0, sizeof(x->pending));
    return C3SC_OK;
}

} // extern "C"
