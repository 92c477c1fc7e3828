// comm_rccl.hip -- the one collective of the sharded Bellman sweep, in C: an all-gather over RCCL (xGMI) on device buffers,
// stream-ordered (north_star: "independent cross-approximation fibers shard across the GPUs of one node with an
// RCCL all-gather").  One process per GPU.  Fibers of a core step are split in contiguous blocks of ceil(F / world) rows, every
// rank evaluates its block on its own device and the blocks are gathered in place (each rank's block already sits at its offset
// of the full array), so that all ranks hold the same F x N values and take the same pivot decisions.
//
// librccl is opened at run time (dlopen), not linked: a process that already carries an RCCL (PyTorch-ROCm bundles one) keeps a
// single instance, and a C program that never shards does not load it at all.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ctx.hpp"

namespace {

typedef int ncclResult;
typedef void *ncclComm;
struct ncclUniqueIdBytes { char internal[128]; }; // NCCL_UNIQUE_ID_BYTES
constexpr int NCCL_FLOAT64 = 8;                   // ncclDataType_t: ncclFloat64

struct Rccl {
    void *lib = nullptr;
    ncclResult (*GetUniqueId)(ncclUniqueIdBytes *) = nullptr;
    ncclResult (*CommInitRank)(ncclComm *, int, ncclUniqueIdBytes, int) = nullptr;
    ncclResult (*CommDestroy)(ncclComm) = nullptr;
    ncclResult (*CommAbort)(ncclComm) = nullptr;
    ncclResult (*AllGather)(const void *, void *, size_t, int, ncclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult) = nullptr;
    std::string err;
};

Rccl &rccl()
{
    static Rccl r;
    if (r.lib || !r.err.empty()) return r;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.err = std::string("librccl not found: ") + dlerror(); return r; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) { r.err = "librccl lacks an expected symbol"; r.lib = nullptr; }
    return r;
}

} // namespace

struct c3sc_hip_comm {
    c3sc_hip_ctx *ctx = nullptr;
    ncclComm comm = nullptr;
    int world = 1, rank = 0;
    double *stage = nullptr; // device staging of the host-buffer exchange (c3sc_hip_comm_exchange)
    size_t stage_doubles = 0;
};

extern "C" {

int c3sc_hip_comm_unique_id(void *id128)
{ // rank 0 calls this and hands the 128 bytes to the other ranks (a file, an environment variable, MPI_Bcast, a socket ...)
    Rccl &r = rccl();
    if (!r.lib || !id128) return C3SC_ERR_ARG;
    ncclUniqueIdBytes id;
    if (r.GetUniqueId(&id) != 0) return C3SC_ERR_HIP;
    std::memcpy(id128, id.internal, 128);
    return C3SC_OK;
}

int c3sc_hip_comm_create(c3sc_hip_ctx *c, int world, int rank, const void *id128, c3sc_hip_comm **out)
{
    if (!c || !out || world < 1 || rank < 0 || rank >= world || !id128) return fail(c, C3SC_ERR_ARG, "comm_create: bad arguments");
    *out = nullptr;
    Rccl &r = rccl();
    if (!r.lib) return fail(c, C3SC_ERR_UNSUPPORTED, r.err.c_str());
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueIdBytes id;
    std::memcpy(id.internal, id128, 128);
    ncclComm comm = nullptr;
    const ncclResult rc = r.CommInitRank(&comm, world, id, rank);
    if (rc != 0) { c->err = std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"); return C3SC_ERR_HIP; }
    c3sc_hip_comm *m = new c3sc_hip_comm();
    m->ctx = c; m->comm = comm; m->world = world; m->rank = rank;
    *out = m;
    return C3SC_OK;
}

void c3sc_hip_comm_destroy(c3sc_hip_comm *m)
{
    if (!m) return;
    if (m->ctx && m->ctx->shard_comm == m) m->ctx->shard_comm = nullptr;
    if (m->stage) (void)hipFree(m->stage);
    if (m->comm) (void)rccl().CommDestroy(m->comm);
    delete m;
}

int c3sc_hip_comm_world(const c3sc_hip_comm *m) { return m ? m->world : 0; }
int c3sc_hip_comm_rank(const c3sc_hip_comm *m) { return m ? m->rank : -1; }

/* every rank contributes count doubles at d_send; d_recv receives world * count doubles in rank order.  In place when
 * d_send == d_recv + rank * count.  Asynchronous on `stream`. */
int c3sc_hip_comm_allgather(c3sc_hip_comm *m, const double *d_send, double *d_recv, size_t count, void *stream)
{
    if (!m || !d_send || !d_recv) return C3SC_ERR_ARG;
    if (!m->comm) { if (m->ctx) m->ctx->err = "comm_allgather: the communicator was aborted after a local failure"; return C3SC_ERR_HIP; }
    Rccl &r = rccl();
    const ncclResult rc = r.AllGather(d_send, d_recv, count, NCCL_FLOAT64, m->comm, (hipStream_t)stream);
    if (rc != 0) { m->ctx->err = std::string("ncclAllGather: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"); return C3SC_ERR_HIP; }
    return C3SC_OK;
}

/* the device-resident cross iterations of ctx (c3sc_hip_cross_iteration[_pi]) shard every core step over the communicator's
 * ranks: rank r evaluates rows [r per, (r + 1) per) of the step's F fibers, per = ceil(F / world), the rows are gathered in
 * place, and memo + factorisation run on the full array on every rank (identical decisions everywhere).  NULL switches it off. */
int c3sc_hip_cross_set_comm(c3sc_hip_ctx *c, c3sc_hip_comm *m)
{
    if (!c) return C3SC_ERR_ARG;
    c->shard_comm = m;
    return C3SC_OK;
}

/* c3sc_exchange_fn of include/c3sc/valuefunc.h for the HOST-driven sharded driver (valuef_interp_idx_sharded,
 * c3control_set_fiber_sharding): xarg = the communicator.  Rank r has filled rows [lo, hi) of the host array out[F][N];
 * afterwards every rank holds all F rows.  The rows travel through a device staging buffer and one ncclAllGather. */
/* A rank that cannot take part in a collective the others are about to enter (or are already waiting in) must not just
 * return: the peers would wait for ever, and a plain C main() has no watchdog.  Where a send buffer still exists the rank
 * enters the all-gather with its rows set to NaN (no fiber value ever is one: every rank sees the mark and all of them
 * return the error together -- the callers check); where it does not (staging allocation failed), the communicator is
 * aborted, which makes the peers' collective return an error instead of blocking. */
static int comm_fail(c3sc_hip_comm *m, const char *what)
{
    if (m->ctx) m->ctx->err = std::string("comm_exchange: ") + what + "; communicator aborted";
    Rccl &r = rccl();
    if (r.CommAbort && m->comm) { (void)r.CommAbort(m->comm); m->comm = nullptr; }
    return 1;
}

int c3sc_hip_comm_exchange(double *out, size_t F, size_t N, size_t lo, size_t hi, void *xarg)
{
    c3sc_hip_comm *m = (c3sc_hip_comm *)xarg;
    if (!m || !out) return 1;
    if (!m->comm) return 1; // aborted earlier: the peers have been released already
    c3sc_hip_ctx *c = m->ctx;
    const size_t per = (F + m->world - 1) / m->world, need = per * m->world * N;
    const bool inject = getenv("C3SC_INJECT_EXCHANGE_FAILURE") != nullptr; // tests: a local failure before the collective
    if (hipSetDevice(c->device) != hipSuccess) return comm_fail(m, "hipSetDevice failed");
    if (need > m->stage_doubles) {
        if (m->stage) (void)hipFree(m->stage);
        m->stage = nullptr; m->stage_doubles = 0;
        if (hipMalloc((void **)&m->stage, need * sizeof(double)) != hipSuccess) return comm_fail(m, "staging buffer allocation failed");
        m->stage_doubles = need;
    }
    double *mine = m->stage + (size_t)m->rank * per * N;
    // local trouble from here on: the rank still ENTERS the collective, with NaN rows (0xFF bytes are a quiet NaN)
    bool bad = inject;
    if (!bad && hipMemsetAsync(mine, 0, per * N * sizeof(double), nullptr) != hipSuccess) bad = true;
    if (!bad && hi > lo && hipMemcpyAsync(mine, out + lo * N, (hi - lo) * N * sizeof(double), hipMemcpyHostToDevice, nullptr) != hipSuccess) bad = true;
    if (bad && hipMemsetAsync(mine, 0xFF, per * N * sizeof(double), nullptr) != hipSuccess) return comm_fail(m, "device unusable");
    if (c3sc_hip_comm_allgather(m, mine, m->stage, per * N, nullptr) != C3SC_OK) return 1; // RCCL itself failed: nobody is waiting on us
    if (hipMemcpyAsync(out, m->stage, F * N * sizeof(double), hipMemcpyDeviceToHost, nullptr) != hipSuccess) return 1;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return 1;
    if (bad) { if (c) c->err = "comm_exchange: local failure before the all-gather (rows sent as NaN)"; return 1; }
    return 0;
}

} // extern "C"
