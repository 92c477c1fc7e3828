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
#include <cstring>

#include "ctx.hpp"

using namespace c3sc;

namespace {

constexpr int NT = 256;                       // threads of the core-step workgroup
constexpr unsigned long long IDX_BITS = 22;   // low bits of a pivot-search key hold the (inverted) index
constexpr unsigned long long IDX_MASK = (1ull << IDX_BITS) - 1;
constexpr size_t LDS_CAP_BYTES = 150 * 1024;  // of the 160 KB of a CU

struct Strides { long long s[MAXD]; };

// ------------------------------------------------------------------------------------------------ fiber index list
// idx[(a + r0 b) d + m] = I_k[a][m] (m < k) | 0 (m = k) | J_k[b][m - k - 1] (m > k)
__global__ void k_cross_idx(int32_t *__restrict__ idx, const int32_t *__restrict__ I, const int32_t *__restrict__ J, int r0, int r1, int k, int d)
{
    const int F = r0 * r1;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < F * d; e += gridDim.x * blockDim.x) {
        const int f = e / d, m = e - f * d;
        const int a = f % r0, b = f / r0;
        int v = 0;
        if (m < k) v = I[a * k + m];
        else if (m > k) v = J[b * (d - 1 - k) + (m - k - 1)];
        idx[e] = v;
    }
}

// ------------------------------------------------------------------------------------------------ node memo
// key word: [63:49] epoch of the sweep (never 0) | [48] pending | [47:0] node id.  A slot whose epoch is not the current one
// is free (the table is cleared by advancing the epoch).  Lookup and insertion in one pass: a hit takes the stored value
// (the reference's memo keeps the first value, bellman.c:1349-1353), a miss stores its own.  The same node twice in ONE
// batch means the same fiber twice (one varying dimension per batch), i.e. identical values: the loser of the race keeps
// its own.
constexpr unsigned long long PENDING = 1ull << 48, EPOCH_MASK = ~((1ull << 49) - 1), ID_MASK = (1ull << 48) - 1;

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
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int inserted = 0;
    if (e < total) {
        const long f = e / N;
        const int j = (int)(e - f * N);
        unsigned long long id = (unsigned long long)j * (unsigned long long)S.s[k];
        for (int m = 0; m < d; m++)
            if (m != k) id += (unsigned long long)idx[f * d + m] * (unsigned long long)S.s[m];
        const unsigned long long K = epoch_bits | id, KP = K | PENDING;
        unsigned long long slot = (id * 0x9E3779B97F4A7C15ull) >> shift;
        const double v = out[e];
        bool done = false;
        for (unsigned long long probe = 0; probe <= capmask && !done; probe++, slot = (slot + 1) & capmask) {
            unsigned long long cur = __hip_atomic_load(&keys[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            while ((cur & EPOCH_MASK) != epoch_bits) { // free: claim it
                const unsigned long long prev = atomicCAS(&keys[slot], cur, KP);
                if (prev == cur) {
                    vals[slot] = v;
                    __hip_atomic_store(&keys[slot], K, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    inserted = 1;
                    done = true;
                    break;
                }
                cur = prev;
            }
            if (done) break;
            if ((cur & ~PENDING) == K) { // this node
                if (!(cur & PENDING)) out[e] = __hip_atomic_load(&vals[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = true;
            }
        }
        if (!done) atomicExch(&counters[3], 1ull); // table full (sized so that it cannot happen)
    }
    const unsigned long long mask = __ballot(inserted);
    if (mask != 0 && (threadIdx.x & 63) == (unsigned)__ffsll((long long)mask) - 1) atomicAdd(&counters[0], (unsigned long long)__popcll(mask));
}

// ------------------------------------------------------------------------------------------------ core step
struct CoreArgs {
    const double *out; // [r0 r1][N] fiber values, fiber f = a + r0 b
    int r0, N, r1, k, d;
    int dir;       // 0: left-to-right (rows (a, j) = a + r0 j, columns b); 1: right-to-left (rows (j, b) = j + N b, columns a)
    int copy_only; // last core of a half sweep: the fiber values are the core
    const int32_t *set_in; // dir 0: I_k [r0][k]; dir 1: J_k [r1][d-1-k]
    int32_t *set_out;      // dir 0: I_{k+1} [r1][k+1]; dir 1: J_{k-1} [r0][d-k]
    double *G;             // core in the working layout G[a + r0 (j + N b)]
    double *work;          // m x n scratch in global memory when the matrix does not fit LDS
    unsigned long long *counters; // [1] rank-deficient factorisation seen, [2] maxvol swaps
};

__device__ inline unsigned long long pivot_key(double x, unsigned long long index)
{ // larger |x| first (22 mantissa bits dropped: values equal to ~2e-10 relative tie), then the LOWER index
    const unsigned long long bits = (unsigned long long)__double_as_longlong(fabs(x));
    return ((bits >> IDX_BITS) << IDX_BITS) | (IDX_MASK - index);
}

__device__ inline unsigned long long block_max(unsigned long long v, unsigned long long *red /* [2][NT / 64] */, int &parity)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    unsigned long long *buf = red + parity * (NT / 64);
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long r = buf[0];
#pragma unroll
    for (int w = 1; w < NT / 64; w++) r = buf[w] > r ? buf[w] : r;
    parity ^= 1; // the next reduction writes the other buffer: one barrier per reduction is enough
    return r;
}

template <bool INLDS>
__global__ void __launch_bounds__(NT) k_cross_core(const CoreArgs P)
{
#pragma clang fp contract(off) // products and sums round separately, as in the host twin (ISO C): the two return the same bits
    extern __shared__ double smem[];
    __shared__ double Lr[32 * 32];
    __shared__ double rowv[32];
    __shared__ double pivabs[32];
    __shared__ int rows[32];
    __shared__ unsigned long long red[2 * (NT / 64)];
    const int tid = threadIdx.x;
    const int r0 = P.r0, N = P.N, r1 = P.r1;
    if (P.copy_only) { // G[a + r0 (j + N b)] = out[(a + r0 b) N + j]
        const long total = (long)r0 * N * r1;
        for (long e = tid; e < total; e += NT) {
            const int a = (int)(e % r0), j = (int)((e / r0) % N), b = (int)(e / ((long)r0 * N));
            P.G[e] = P.out[((long)a + (long)r0 * b) * N + j];
        }
        return;
    }
    const int m = P.dir == 0 ? r0 * N : N * r1, n = P.dir == 0 ? r1 : r0;
    double *A = INLDS ? smem : P.work; // column-major m x n
    // load the fiber matrix
    for (long e = tid; e < (long)m * n; e += NT) {
        const int i = (int)(e % m), c = (int)(e / m);
        int a, j, b;
        if (P.dir == 0) { a = i % r0; j = i / r0; b = c; }
        else { j = i % N; b = i / N; a = c; }
        A[e] = P.out[((long)a + (long)r0 * b) * N + j];
    }
    __syncthreads();
    int parity = 0;
    unsigned used = 0; // bit q: my q-th row (row tid + q NT) is a pivot row
    // ---- tall LU with row pivoting: A = P L U, |L| <= 1; afterwards A holds L below the pivots (multipliers in place)
    for (int kc = 0; kc < n; kc++) {
        unsigned long long key = 0;
        for (int i = tid, q = 0; i < m; i += NT, q++)
            if (!((used >> q) & 1u)) {
                const unsigned long long kk = pivot_key(A[i + (long)kc * m], (unsigned long long)i);
                key = kk > key ? kk : key;
            }
        const unsigned long long best = block_max(key, red, parity);
        const int p = (int)(IDX_MASK - (best & IDX_MASK));
        const double dp = A[p + (long)kc * m];
        if (tid == 0) { rows[kc] = p; pivabs[kc] = fabs(dp); }
        if (p % NT == tid) used |= 1u << (p / NT);
        const double inv = dp != 0.0 ? 1.0 / dp : 0.0;
        for (int i = tid, q = 0; i < m; i += NT, q++) {
            if ((used >> q) & 1u) continue;
            const double l = A[i + (long)kc * m] * inv;
            A[i + (long)kc * m] = l;
            if (l != 0.0)
                for (int c = kc + 1; c < n; c++) A[i + (long)c * m] -= l * A[p + (long)c * m];
        }
        __syncthreads();
    }
    // ---- B = L inv(L[rows]): L[rows] is unit lower triangular in pivot order
    for (int e = tid; e < n * n; e += NT) {
        const int q = e / n, j = e % n;
        Lr[q * 32 + j] = j < q ? A[rows[q] + (long)j * m] : (j == q ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int i = tid, q = 0; i < m; i += NT, q++) {
        if ((used >> q) & 1u) continue;
        for (int j = n - 1; j >= 0; j--) { // x Lr = l, in place
            double s = A[i + (long)j * m];
            for (int t = j + 1; t < n; t++) s -= A[i + (long)t * m] * Lr[t * 32 + j];
            A[i + (long)j * m] = s;
        }
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += NT) { // pivot rows: unit vectors
        const int q = e / n, j = e % n;
        A[rows[q] + (long)j * m] = (j == q) ? 1.0 : 0.0;
    }
    __syncthreads();
    // ---- maxvol: swap rows until the largest entry of B is <= 1.01
    int nswaps = 0;
    for (int it = 0; it < 200; it++) {
        unsigned long long key = 0;
        for (int c = 0; c < n; c++)
            for (int i = tid; i < m; i += NT) {
                const unsigned long long kk = pivot_key(A[i + (long)c * m], (unsigned long long)c * (unsigned long long)m + (unsigned long long)i);
                key = kk > key ? kk : key;
            }
        const unsigned long long best = block_max(key, red, parity);
        const long lin = (long)(IDX_MASK - (best & IDX_MASK));
        const int bj = (int)(lin / m), bi = (int)(lin % m);
        const double piv = A[bi + (long)bj * m];
        if (!(fabs(piv) > 1.0 + 1e-2)) break;
        if (tid < n) rowv[tid] = A[bi + (long)tid * m] - (tid == bj ? 1.0 : 0.0);
        __syncthreads();
        for (int i = tid; i < m; i += NT) {
            const double cv = A[i + (long)bj * m] / piv;
            if (cv != 0.0)
                for (int c = 0; c < n; c++) A[i + (long)c * m] -= cv * rowv[c];
        }
        if (tid == 0) rows[bj] = bi;
        nswaps++;
        __syncthreads();
    }
    __syncthreads();
    // ---- results
    if (P.dir == 0) {
        for (long e = tid; e < (long)m * n; e += NT) P.G[e] = A[e]; // G[i + m b], i = a + r0 j
        const int k = P.k;
        for (int e = tid; e < n * (k + 1); e += NT) {
            const int q = e / (k + 1), t = e % (k + 1);
            const int row = rows[q], a = row % r0, j = row / r0;
            P.set_out[e] = t < k ? P.set_in[a * k + t] : j;
        }
    } else {
        for (long e = tid; e < (long)m * n; e += NT) { // G[a + r0 cc] = B[cc][a]
            const int a = (int)(e % r0);
            const long cc = e / r0;
            P.G[e] = A[cc + (long)a * m];
        }
        const int len = P.d - P.k;
        for (int e = tid; e < n * len; e += NT) {
            const int q = e / len, t = e % len;
            const int row = rows[q], j = row % N, b = row / N;
            P.set_out[e] = t == 0 ? j : P.set_in[b * (len - 1) + (t - 1)];
        }
    }
    if (tid == 0) {
        double mx = 0.0, mn = INFINITY;
        for (int c = 0; c < n; c++) { mx = pivabs[c] > mx ? pivabs[c] : mx; mn = pivabs[c] < mn ? pivabs[c] : mn; }
        if (!(mn > 1e-12 * mx)) atomicExch(&P.counters[1], 1ull);
        if (nswaps) atomicAdd(&P.counters[2], (unsigned long long)nswaps);
    }
}

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
    // memo
    unsigned long long *keys = nullptr;
    double *vals = nullptr;
    size_t cap = 0;
    unsigned epoch = 0;
    Strides strides;
    unsigned long long *counters = nullptr; // device [4]
    // pinned host staging for the one-copy upload / download
    char *stage = nullptr;
    size_t stage_bytes = 0;
    bool lds_optin = false;
};

static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" {

void c3sc_hip_cross_free(c3sc_hip_ctx *c)
{
    if (!c || !c->cross) return;
    c3sc_cross_dev *x = c->cross;
    if (x->slab) (void)hipFree(x->slab);
    if (x->keys) (void)hipFree(x->keys);
    if (x->vals) (void)hipFree(x->vals);
    if (x->counters) (void)hipFree(x->counters);
    if (x->stage) (void)hipHostFree(x->stage);
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
        if (ranks[k] < 1 || ranks[k] > 32) return fail(c, C3SC_ERR_UNSUPPORTED, "cross_setup: ranks up to 32");
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
        if (sz > (IDX_MASK + 1) / 2 || (size_t)x->r[k] * x->N[k] > 32 * NT || (size_t)x->N[k] * x->r[k + 1] > 32 * NT)
            return fail(c, C3SC_ERR_UNSUPPORTED, "cross_setup: core too large for the one-workgroup factorisation");
        x->offG[k] = off; off += up256(sz * sizeof(double));
    }
    x->cores_bytes = off - x->off_cores;
    x->off_idx = off; off += up256(fmax * d * sizeof(int32_t));
    x->off_out = off; off += up256(fmax * nmax * sizeof(double));
    x->off_work = off; off += up256(wmax * sizeof(double));
    if (off > x->slab_bytes) {
        if (x->slab) HIPCHK(c, hipFree(x->slab));
        x->slab = nullptr; x->slab_bytes = 0;
        HIPCHK(c, hipMalloc((void **)&x->slab, off));
        x->slab_bytes = off;
    }
    const size_t need_stage = x->sets_bytes + x->cores_bytes + 64;
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
    // memo: every node a sweep can touch (5 iterations x 2 half sweeps x sum_k F_k N_k, before rank kicks) at load <= 1/2
    size_t nodes = 0;
    for (int k = 0; k < d; k++) nodes += (size_t)x->r[k] * x->r[k + 1] * x->N[k];
    size_t want = 1 << 16;
    while (want < 32 * nodes) want <<= 1;
    if (want > x->cap) { // grow; a sweep in progress (rank kick between cross rounds) keeps its entries
        unsigned long long *nk = nullptr;
        double *nv = nullptr;
        HIPCHK(c, hipMalloc((void **)&nk, want * sizeof(unsigned long long)));
        HIPCHK(c, hipMalloc((void **)&nv, want * sizeof(double)));
        HIPCHK(c, hipMemset(nk, 0, want * sizeof(unsigned long long)));
        if (x->keys && !new_sweep && x->epoch != 0) {
            int sh = 64;
            for (size_t cp = want; cp > 1; cp >>= 1) sh--;
            hipLaunchKernelGGL(k_cross_memo_rehash, dim3(256), dim3(256), 0, nullptr, x->keys, x->vals, (unsigned long long)x->cap, nk, nv,
                               (unsigned long long)(want - 1), sh, (unsigned long long)x->epoch << 49);
            HIPCHK(c, hipDeviceSynchronize());
        } else {
            x->epoch = 0;
        }
        if (x->keys) HIPCHK(c, hipFree(x->keys));
        if (x->vals) HIPCHK(c, hipFree(x->vals));
        x->keys = nk; x->vals = nv; x->cap = want;
    }
    if (new_sweep || x->epoch == 0) {
        x->epoch++;
        if (x->epoch > 0x7FFF) { // epoch field wrapped: really clear
            HIPCHK(c, hipMemset(x->keys, 0, x->cap * sizeof(unsigned long long)));
            x->epoch = 1;
        }
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
        x->lds_optin = true;
    }
    return C3SC_OK;
}

/* one cross iteration: left-to-right half sweep (new left index sets), then right-to-left (new right index sets); the cores
 * of the right-to-left half sweep are the iteration's result */
int c3sc_hip_cross_iteration(c3sc_hip_ctx *c, int box, void *stream)
{
    if (!c || !c->cross) return fail(c, C3SC_ERR_ARG, "cross_iteration: cross_setup first");
    c3sc_cross_dev *x = c->cross;
    const int d = x->d;
    hipStream_t st = (hipStream_t)stream;
    int shift = 64;
    for (size_t cp = x->cap; cp > 1; cp >>= 1) shift--;
    for (int half = 0; half < 2; half++)
        for (int s = 0; s < d; s++) {
            const int k = half == 0 ? s : d - 1 - s;
            const int r0 = x->r[k], r1 = x->r[k + 1], N = x->N[k];
            const size_t F = (size_t)r0 * r1;
            int32_t *idx = (int32_t *)(x->slab + x->off_idx);
            double *out = (double *)(x->slab + x->off_out);
            const int32_t *Ik = (const int32_t *)(x->slab + x->offI[k]), *Jk = (const int32_t *)(x->slab + x->offJ[k]);
            hipLaunchKernelGGL(k_cross_idx, dim3((unsigned)((F * d + 255) / 256)), dim3(256), 0, st, idx, Ik, Jk, r0, r1, k, d);
            int rc = box ? c3sc_hip_bellman_fibers_box(c, k, F, idx, out, nullptr, nullptr, stream)
                         : c3sc_hip_bellman_fibers(c, k, F, idx, out, nullptr, nullptr, stream);
            if (rc != C3SC_OK) return rc;
            const long total = (long)F * N;
            hipLaunchKernelGGL(k_cross_memo, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, idx, out, total, N, d, k, x->strides,
                               x->keys, x->vals, (unsigned long long)(x->cap - 1), shift, (unsigned long long)x->epoch << 49, x->counters);
            CoreArgs P;
            P.out = out; P.r0 = r0; P.N = N; P.r1 = r1; P.k = k; P.d = d;
            P.dir = half;
            P.copy_only = (half == 0) ? (k == d - 1) : (k == 0);
            P.set_in = half == 0 ? Ik : Jk;
            P.set_out = P.copy_only ? nullptr : (half == 0 ? (int32_t *)(x->slab + x->offI[k + 1]) : (int32_t *)(x->slab + x->offJ[k - 1]));
            P.G = (double *)(x->slab + x->offG[k]);
            P.work = (double *)(x->slab + x->off_work);
            P.counters = x->counters;
            const size_t mn = F * N * sizeof(double);
            if (P.copy_only || mn <= LDS_CAP_BYTES)
                hipLaunchKernelGGL(k_cross_core<true>, dim3(1), dim3(NT), P.copy_only ? 0 : mn, st, P);
            else
                hipLaunchKernelGGL(k_cross_core<false>, dim3(1), dim3(NT), 0, st, P);
        }
    HIPCHK(c, hipGetLastError());
    return C3SC_OK;
}

/* wait for the iteration and bring back: cores (working layout G[a + r_k (j + N_k b)]), both families of index sets, and
 * info = {nodes stored in the memo since the last fetch, rank-deficient factorisation seen, maxvol swaps, memo overflow} */
int c3sc_hip_cross_fetch(c3sc_hip_ctx *c, double *const *h_cores, int32_t *const *h_I, int32_t *const *h_J, unsigned long long *info,
                         void *stream)
{
    if (!c || !c->cross) return fail(c, C3SC_ERR_ARG, "cross_fetch: cross_setup first");
    c3sc_cross_dev *x = c->cross;
    hipStream_t st = (hipStream_t)stream;
    const size_t nb = x->sets_bytes + x->cores_bytes;
    HIPCHK(c, hipMemcpyAsync(x->stage, x->slab, nb, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(x->stage + nb, x->counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemsetAsync(x->counters, 0, 4 * sizeof(unsigned long long), st));
    HIPCHK(c, hipStreamSynchronize(st));
    const int d = x->d;
    for (int k = 0; k < d; k++) {
        if (h_cores) std::memcpy(h_cores[k], x->stage + x->offG[k], (size_t)x->r[k] * x->N[k] * x->r[k + 1] * sizeof(double));
        if (h_I) std::memcpy(h_I[k], x->stage + x->offI[k], (size_t)x->r[k] * k * sizeof(int32_t));
        if (h_J) std::memcpy(h_J[k], x->stage + x->offJ[k], (size_t)x->r[k + 1] * (d - 1 - k) * sizeof(int32_t));
    }
    if (info) std::memcpy(info, x->stage + nb, 4 * sizeof(unsigned long long));
    return C3SC_OK;
}

} // extern "C"
