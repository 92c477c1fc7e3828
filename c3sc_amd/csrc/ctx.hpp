// ctx.hpp -- the device context behind the C-ABI handle (shared by c3sc_hip.hip and cross_device.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/c3sc_hip.h"
#include "kernel_common.hpp"

struct c3sc_cross_dev; // cross_device.hip: device-resident state of the cross-approximation core steps
struct c3sc_hip_comm;  // comm_rccl.hip: RCCL communicator (one process per GPU)

struct c3sc_hip_ctx {
    int device = 0;
    std::string err;
    // host copy of the static problem description
    int d = 0;
    int ngrid[c3sc::MAXD] = {0};
    std::vector<double> xgrid_flat;
    int xg_off_rel[c3sc::MAXD] = {0};
    int bctype[c3sc::MAXD] = {0};
    int cends = 0; // c3sc_hip_set_consistent_ends
    bool have_boundary = false;
    int nobs = 0;
    std::vector<double> obs; // [nobs][2][d]
    bool have_mca = false;
    double h2 = 0, discount = 0, t[2 * c3sc::MAXD] = {0};
    int model = 0;
    double prm[C3SC_MAX_PARAMS] = {0};
    int ncand = 0, du = 0;
    std::vector<double> cands;
    // continuous controls (c3sc_hip_set_control_box)
    int box_du = 0, box_grid = 0, box_polish = 0;
    double box_lb[C3SC_MAX_DU] = {0}, box_ub[C3SC_MAX_DU] = {0};
    // value function
    bool have_value = false;
    size_t ranks[c3sc::MAXD + 1] = {0};
    int rp = 0;
    // device arena: [xgrid | obs | cands | pad | cores]
    double *arena = nullptr;
    size_t arena_cap = 0;     // doubles
    size_t static_doubles = 0; // size of the static section the arena was laid out with
    bool static_dirty = true;
    long core_off[c3sc::MAXD] = {0};
    long coreT_off[c3sc::MAXD] = {0}, aop_off[c3sc::MAXD] = {0}; // fiber-quad copies of the middle cores (0 = none)
    long img_off[c3sc::MAXD] = {0};                        // fiber-pair LDS images of all cores (padded node stride)
    long qimgL_off[c3sc::MAXD] = {0}, qimgR_off[c3sc::MAXD] = {0}; // fiber-quad-duo LDS images (node stride elems + 2)
    int obs_off = 0, cands_off = 0, tab_off[4] = {0, 0, 0, 0}, cfeat_off = 0;
    unsigned *d_status = nullptr;
    // the status word as of the last c3sc_hip_cross_fetch, valid until the next launch through this context: the solver reads the
    // status once per sweep, right after the fetch that already waited for the stream -- no second device round trip
    unsigned status_cache = 0;
    bool status_cache_valid = false;
    unsigned long long *d_dbg = nullptr; // diagnostic stamps (C3SC_DBG & 128)
    int variant = C3SC_VARIANT_AUTO;
    const char *last_kernel = "";
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // c3sc_hip_bellman_fibers_all: side streams the independent per-dimension launches of one batch are spread over, with the
    // events that fork them from and join them to the caller's stream (created on first use)
    static constexpr int NSIDE = 2;
    hipStream_t side[NSIDE] = {nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[NSIDE] = {nullptr, nullptr};
    // scratch for the *_host convenience calls
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // pinned, device-mapped host block for small *_host batches (read and written by the kernel in place)
    void *pinned = nullptr, *pinned_dev = nullptr;
    size_t pinned_bytes = 0;
    // node memo requested for the next Bellman launch (set by c3sc_hip_cross_iteration around its launches)
    struct {
        unsigned long long *keys = nullptr;
        double *vals = nullptr;
        unsigned long long capmask = 0, epoch_bits = 0;
        int shift = 0;
        long long stride[c3sc::MAXD] = {0};
        unsigned long long *counters = nullptr;
        int mode = 0;         // KArgs::memo_mode
        bool applied = false; // the launched kernel carried the memo in its epilogue
    } memo;
    const int *skip_flag = nullptr; // device flag the next fiber-per-wave launch returns on at once when set (cross_device.hip)
    c3sc_hip_comm *shard_comm = nullptr; // borrowed: the cross iterations shard their core steps over it (c3sc_hip_cross_set_comm)
    c3sc_cross_dev *cross = nullptr; // owned; freed by c3sc_hip_cross_free (called from c3sc_hip_ctx_destroy)
};

#define HIPCHK(ctx, call)                                                                    \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess) {                                                             \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                 \
            return C3SC_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

static inline int fail(c3sc_hip_ctx *ctx, int code, const char *msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

