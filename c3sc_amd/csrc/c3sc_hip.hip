// c3sc_hip.hip -- host side of libc3sc_hip.so: the C-ABI declared in include/c3sc_hip.h.
// Owns the device-resident problem description (one read-only arena: grids, obstacles, control
// candidates, rank-padded FT cores), selects a kernel instantiation and launches it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/c3sc_hip.h"
#include "kernel_common.hpp"
#include "registry.hpp"

namespace c3sc {

std::vector<KernelEntry> &kernel_registry()
{
    static std::vector<KernelEntry> reg;
    return reg;
}

// Re-pack one FT core from the reference layout cores[m][j*r0*r1 + a + b*r0] (valuefunc.c:165-189)
// into the rank-padded device layout: first core [N][RP] (index b), last core [N][RP] (index a),
// middle cores [N][RP*RP] (a + b*RP); padding entries are zero, which leaves every contraction exact.
__global__ void k_pad_core(const double *__restrict__ src, double *__restrict__ dst, int N, int r0, int r1, int RP,
                           int kind /*0 first, 1 middle, 2 last*/)
{
    const int per = (kind == 1) ? RP * RP : RP;
    const long total = (long)N * per;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e / per), w = (int)(e - (long)j * per);
        int a, b;
        if (kind == 0) { a = 0; b = w; }
        else if (kind == 2) { a = w; b = 0; }
        else { a = w % RP; b = w / RP; }
        double v = 0.0;
        if (a < r0 && b < r1) v = src[(size_t)j * r0 * r1 + a + (size_t)b * r0];
        dst[e] = v;
    }
}

// LDS images of the cores (the arena layout once more with the node stride a kernel uses in LDS), all in ONE launch: the
// fiber-pair kernel copies its fixed cores into LDS by LDS-DMA (global_load_lds, 16 bytes per lane), which writes
// lane-linearly and cannot pad, so the padding (elems | 1 doubles per node there, elems + 2 for the fiber-quad kernels) is
// laid down here.  blockIdx.y = job.
struct ImgJobs {
    int n;
    long src[3 * MAXD], dst[3 * MAXD];
    int nodes[3 * MAXD], per[3 * MAXD], stride[3 * MAXD];
};
__global__ void k_core_images(double *__restrict__ arena, const ImgJobs J)
{
    const int jb = blockIdx.y;
    const double *core = arena + J.src[jb];
    double *img = arena + J.dst[jb];
    const int per = J.per[jb], stride = J.stride[jb];
    const long total = (long)J.nodes[jb] * stride;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e / stride), w = (int)(e - (long)j * stride);
        img[e] = (w < per) ? core[(size_t)j * per + w] : 0.0;
    }
}

// Derived copies of a rank-padded middle core for the fiber-quad kernel (kernel_fiber_quad.hpp), made on the device from
// the padded core itself: (1) the row-major transpose, (2) the two MFMA A operands of the varying-core products
// c = G R (x = a, y = b) and a = L G (x = b, y = a): element [prod][mb][s][l] = M[x][y] with x = (i%4) C + 4 mb + i/4
// (i = l % 16; a zero row when 4 mb + i/4 >= C) and y = (l/16) C + s, so that D register r of lane (q, t) is component
// q C + 4 mb + r of the product for fiber t.
__global__ void k_quad_aux(const double *__restrict__ core, double *__restrict__ coreT, double *__restrict__ aop, int N, int RP)
{
    const int C = RP / 4, MB = (C + 3) / 4;
    const int per_t = RP * RP, per_a = 2 * MB * C * 64;
    const long total = (long)N * (per_t + per_a);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e / (per_t + per_a)), w = (int)(e - (long)j * (per_t + per_a));
        const double *G = core + (size_t)j * per_t; // a + b*RP
        if (w < per_t) {
            const int a = w / RP, b = w % RP;
            coreT[(size_t)j * per_t + w] = G[a + b * RP];
        } else {
            const int u = w - per_t;
            const int l = u % 64, s = (u / 64) % C, mb = (u / (64 * C)) % MB, prod = u / (64 * C * MB);
            const int i = l % 16, g = 4 * mb + i / 4; // D register r of lane (q, t) is row 4 r + q of the 16 x 16 block (probed:
            const int xc = (i % 4) * C + g, yc = (l / 16) * C + s; // tools/probe_mfma_layout.hip), so A row i serves (q, r) = (i % 4, i / 4)
            double v = 0.0;
            if (g < C) v = (prod == 0) ? G[xc + yc * RP] : G[yc + xc * RP];
            aop[(size_t)j * per_a + u] = v;
        }
    }
}

// FP64 peak probes (DESIGN.md "Roofline peaks"): dependent-free FMA streams per lane
__global__ void k_peak_fma(double *out, int iters)
{
    double a0 = threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    for (int i = 0; i < iters; i++) {
        a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
        a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k_peak_mfma(double *out, int iters)
{
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

} // namespace c3sc

using namespace c3sc;

#include "ctx.hpp"

// Host companions of the device models (models.hpp): the univariate functions of a grid coordinate and of
// a control candidate are evaluated HERE with libm, exactly as the reference's callbacks would
// (dubinscar.c:48-49, scar.c:65-67), and handed to the kernels as tables.
// Besides the trigonometric tables, any other expensive univariate function of a grid coordinate is tabulated
// here with the host's IEEE arithmetic (correctly rounded division, the same result the device's division sequence
// gives): the car models' speed factor v / (0.2 (1 + v/8)) costs a ~15-instruction dependent chain per node otherwise.
static int model_ntab(int model)
{
    if (model == C3SC_MODEL_SKID5D) return 2; // cos / sin of the orientation x2
    return model == C3SC_MODEL_DUBINS3D ? 2 : (model == C3SC_MODEL_SCAR4D ? 3 : ((model == C3SC_MODEL_CAR7D || model == C3SC_MODEL_PERCH7D) ? 4 : 0));
}
static int model_tab_dim(int model, int t)
{
    if (model == C3SC_MODEL_PERCH7D) return t < 2 ? 2 : 3; // cos / sin of the pitch x2, cos / sin of the elevator angle x3
    if (model == C3SC_MODEL_CAR7D) return t == 2 ? 5 : (t == 3 ? 3 : 2);
    if (model == C3SC_MODEL_SCAR4D) return t == 2 ? 3 : 2;
    return 2;
}
static double model_table_value(int model, int t, double xv)
{
    if (t == 0) return cos(xv);
    if (t == 1) return sin(xv);
    if (model == C3SC_MODEL_PERCH7D) return t == 2 ? cos(xv) : sin(xv);
    if (model == C3SC_MODEL_CAR7D && t == 2) return tan(xv);
    if (model == C3SC_MODEL_CAR7D && t == 3) return xv / (0.2 * (1.0 + xv / 8.0));
    if (model == C3SC_MODEL_SCAR4D && t == 2) return (1.0 / (1.0 + (xv / 8.0))) * (xv / 0.2); /* scar.c:68-71 with L = 0.2, vcar = 8 */
    return tan(xv);
}
static int model_ncf(int model) { return model == C3SC_MODEL_SCAR4D ? 1 : (model == C3SC_MODEL_COTHRUST6D ? 3 : 0); }
static double model_cand_feature(int model, int q, const double *u)
{
    if (model == C3SC_MODEL_COTHRUST6D) { /* copterposethrust.c:104-117, the callback's own expressions */
        const double m = 1.227, g = 9.81, mg = m * g;
        const double cphi = cos(u[1]), sphi = sin(u[1]), cth = cos(u[2]), sth = sin(u[2]);
        return q == 0 ? cphi * sth * (u[0] - mg) / m : (q == 1 ? -sphi * (u[0] - mg) / m : g + cth * cphi * (u[0] - mg) / m);
    }
    (void)q;
    return tan(u[0]); /* scar.c:67 */
}

static size_t static_layout(c3sc_hip_ctx *c)
{ // offsets of the static section; returns its size in doubles (rounded to 16)
    size_t off = c->xgrid_flat.size();
    c->obs_off = (int)off;
    off += c->obs.size();
    c->cands_off = (int)off;
    off += c->cands.size();
    for (int t = 0; t < 4; t++) c->tab_off[t] = 0;
    const int nt = model_ntab(c->model);
    for (int t = 0; t < nt; t++) {
        const int dim = model_tab_dim(c->model, t);
        c->tab_off[t] = (int)off;
        off += (dim < c->d) ? (size_t)c->ngrid[dim] : 0;
    }
    c->cfeat_off = (int)off;
    off += (size_t)model_ncf(c->model) * c->ncand;
    return (off + 15) & ~(size_t)15;
}

static int upload_static(c3sc_hip_ctx *c)
{
    std::vector<double> st(c->static_doubles, 0.0);
    std::copy(c->xgrid_flat.begin(), c->xgrid_flat.end(), st.begin());
    std::copy(c->obs.begin(), c->obs.end(), st.begin() + c->obs_off);
    std::copy(c->cands.begin(), c->cands.end(), st.begin() + c->cands_off);
    const int nt = model_ntab(c->model);
    for (int t = 0; t < nt; t++) {
        const int dim = model_tab_dim(c->model, t);
        if (dim >= c->d) continue;
        const double *g = c->xgrid_flat.data() + c->xg_off_rel[dim];
        for (int i = 0; i < c->ngrid[dim]; i++) st[c->tab_off[t] + i] = model_table_value(c->model, t, g[i]);
    }
    const int ncf = model_ncf(c->model);
    for (int q = 0; q < c->ncand * ncf; q++)
        st[c->cfeat_off + q] = model_cand_feature(c->model, q % ncf, c->cands.data() + (size_t)(q / ncf) * c->du);
    HIPCHK(c, hipMemcpy(c->arena, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice));
    c->static_dirty = false;
    return C3SC_OK;
}

// AUTO picks by batch size: the fiber-pair kernel needs 64 fibers x 1024 resident workgroups to fill the chip and a
// single tile takes ~0.12 ms whatever F is, while one-wave-per-fiber scales down to ~20 us (measured on car7d:
// 0.020 / 0.065 / 0.23 ms at F = 2k / 8k / 32k against 0.12 / 0.12 / 0.15 ms) -- cross-approximation core steps
// (F = r_k r_{k+1}, a few hundred fibers) are latency-bound and take the per-wave kernel.
static const size_t SMALL_BATCH_FIBERS = 16384;

static const KernelEntry *find_kernel(int model, int d, int rank_needed, int N, int variant, int k, size_t F = (size_t)-1,
                                      const std::vector<const KernelEntry *> *skip = nullptr)
{
    const KernelEntry *best = nullptr;
    const bool small = (variant == C3SC_VARIANT_AUTO) && F < SMALL_BATCH_FIBERS;
    for (const auto &e : kernel_registry()) {
        if (e.model != model || e.d != d || e.rp < rank_needed || e.max_n < N) continue;
        if (skip && std::find(skip->begin(), skip->end(), &e) != skip->end()) continue;
        if (e.k >= 0 && e.k != k) continue;
        if (variant != C3SC_VARIANT_AUTO && e.variant != variant) continue;
        auto pref = [small](int v) {
            if (small) return v == C3SC_VARIANT_FIBER_PER_WAVE ? 0 : (v == C3SC_VARIANT_FIBER_PAIR ? 1 : (v == C3SC_VARIANT_FIBER_QUAD ? 2 : 3));
            return v == C3SC_VARIANT_FIBER_PAIR ? 0 : (v == C3SC_VARIANT_FIBER_QUAD ? 1 : (v == C3SC_VARIANT_FIBER_PER_WAVE ? 2 : 3));
        };
        if (!best || e.rp < best->rp || (e.rp == best->rp && pref(e.variant) < pref(best->variant)) ||
            (e.rp == best->rp && e.variant == best->variant && e.npl < best->npl))
            best = &e;
    }
    return best;
}

static int pick_rp(int d, int maxrank, int model, int variant)
{ // smallest padded rank an instantiation of this dimension offers -- of the selected variant and model when a variant
  // is forced (its padded ranks may differ from the other kernels': the quad kernel wants multiples of 4)
    int rp = 0;
    if (variant != C3SC_VARIANT_AUTO)
        for (const auto &e : kernel_registry())
            if (e.d == d && e.variant == variant && (model == 0 || e.model == model) && e.rp >= maxrank && (rp == 0 || e.rp < rp)) rp = e.rp;
    if (rp) return rp;
    if (model != 0) // the padded classes compiled for THIS model (another model of the same dimension may offer others)
        for (const auto &e : kernel_registry())
            if (e.d == d && e.model == model && e.rp >= maxrank && (rp == 0 || e.rp < rp)) rp = e.rp;
    if (rp) return rp;
    for (const auto &e : kernel_registry())
        if (e.d == d && e.rp >= maxrank && (rp == 0 || e.rp < rp)) rp = e.rp;
    return rp;
}

static unsigned long long g_launches = 0; // Bellman / stencil kernel launches of this process (c3sc_hip_launch_count)
static int ensure_scratch(c3sc_hip_ctx *c, size_t bytes);
static int ensure_pinned(c3sc_hip_ctx *c, size_t bytes);
static bool zero_copy_batch(size_t bytes);
static size_t align256(size_t x);
static int fill_args(c3sc_hip_ctx *c, int k, size_t F, KArgs &A, bool need_model);

extern "C" {

int c3sc_hip_max_rank(int model, int d)
{ // largest FT rank any compiled kernel of this (model, state dimension) serves; 0 = none
    int rp = 0;
    for (const auto &e : kernel_registry())
        if (e.model == model && e.d == d && e.rp > rp) rp = e.rp;
    return rp;
}

int c3sc_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int c3sc_hip_ctx_create(int device, c3sc_hip_ctx **out)
{
    if (!out) return C3SC_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return C3SC_ERR_NODEVICE;
    c3sc_hip_ctx *c = new c3sc_hip_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&c->d_status, sizeof(unsigned)) != hipSuccess ||
        hipMemset(c->d_status, 0, sizeof(unsigned)) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return C3SC_ERR_HIP;
    }
    *out = c;
    return C3SC_OK;
}

void c3sc_hip_ctx_destroy(c3sc_hip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->arena) (void)hipFree(c->arena);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->d_dbg) (void)hipFree(c->d_dbg);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->pinned) (void)hipHostFree(c->pinned);
    c3sc_hip_cross_free(c);
    for (int i = 0; i < c3sc_hip_ctx::NSIDE; i++) {
        if (c->side[i]) { (void)hipStreamSynchronize(c->side[i]); (void)hipStreamDestroy(c->side[i]); }
        if (c->join_ev[i]) (void)hipEventDestroy(c->join_ev[i]);
    }
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    delete c;
}

const char *c3sc_hip_last_error(const c3sc_hip_ctx *c) { return c ? c->err.c_str() : "null context"; }
const char *c3sc_hip_last_kernel(const c3sc_hip_ctx *c) { return c ? c->last_kernel : ""; }

int c3sc_hip_set_grid(c3sc_hip_ctx *c, int d, const size_t *ngrid, const double *const *xgrid)
{
    if (!c || !ngrid || !xgrid || d < 2 || d > MAXD) return fail(c, C3SC_ERR_ARG, "set_grid: need 2 <= d <= 12");
    c->d = d;
    c->xgrid_flat.clear();
    for (int m = 0; m < d; m++) {
        if (ngrid[m] < 2 || ngrid[m] > 4096) return fail(c, C3SC_ERR_ARG, "set_grid: need 2 <= N_m <= 4096");
        c->ngrid[m] = (int)ngrid[m];
        c->xg_off_rel[m] = (int)c->xgrid_flat.size();
        c->xgrid_flat.insert(c->xgrid_flat.end(), xgrid[m], xgrid[m] + ngrid[m]);
    }
    c->static_dirty = true;
    c->have_value = false; // grid change invalidates the uploaded value function
    return C3SC_OK;
}

int c3sc_hip_set_boundary(c3sc_hip_ctx *c, const int *bctype, int nobs, const double *obs_lb, const double *obs_ub)
{
    if (!c || c->d == 0 || !bctype) return fail(c, C3SC_ERR_ARG, "set_boundary: set_grid first");
    if (nobs < 0 || nobs > C3SC_MAX_OBSTACLES) return fail(c, C3SC_ERR_ARG, "set_boundary: at most 10 obstacles (boundary.c:393)");
    for (int m = 0; m < c->d; m++) {
        if (bctype[m] != C3SC_ABSORB && bctype[m] != C3SC_PERIODIC && bctype[m] != C3SC_REFLECT)
            return fail(c, C3SC_ERR_ARG, "set_boundary: boundary type must be absorb/periodic/reflect (nodeutil.c:532-535)");
        c->bctype[m] = bctype[m];
    }
    c->nobs = nobs;
    c->obs.assign((size_t)nobs * 2 * c->d, 0.0);
    for (int o = 0; o < nobs; o++)
        for (int m = 0; m < c->d; m++) {
            c->obs[((size_t)o * 2 + 0) * c->d + m] = obs_lb[(size_t)o * c->d + m];
            c->obs[((size_t)o * 2 + 1) * c->d + m] = obs_ub[(size_t)o * c->d + m];
        }
    c->have_boundary = true;
    c->static_dirty = true;
    return C3SC_OK;
}

int c3sc_hip_set_consistent_ends(c3sc_hip_ctx *c, int on)
{
    if (!c) return C3SC_ERR_ARG;
    c->cends = on ? 1 : 0;
    return C3SC_OK;
}

int c3sc_hip_get_consistent_ends(const c3sc_hip_ctx *c) { return c ? c->cends : -1; }

int c3sc_hip_set_mca(c3sc_hip_ctx *c, double h2, const double *t, double discount)
{
    if (!c || c->d == 0 || !t) return fail(c, C3SC_ERR_ARG, "set_mca: set_grid first");
    c->h2 = h2;
    c->discount = discount;
    for (int i = 0; i < 2 * c->d; i++) c->t[i] = t[i];
    c->have_mca = true;
    return C3SC_OK;
}

int c3sc_hip_set_model(c3sc_hip_ctx *c, int model, const double *params, int nparams)
{
    if (!c || model <= 0 || nparams < 0 || nparams > C3SC_MAX_PARAMS) return fail(c, C3SC_ERR_ARG, "set_model: bad arguments");
    c->model = model;
    std::memset(c->prm, 0, sizeof(c->prm));
    for (int i = 0; i < nparams; i++) c->prm[i] = params[i];
    c->static_dirty = true; // model tables live in the static section
    return C3SC_OK;
}

int c3sc_hip_set_controls(c3sc_hip_ctx *c, int ncand, int du, const double *cands)
{
    if (!c || ncand < 1 || du < 1 || !cands) return fail(c, C3SC_ERR_ARG, "set_controls: bad arguments");
    c->ncand = ncand;
    c->du = du;
    c->cands.assign(cands, cands + (size_t)ncand * du);
    c->static_dirty = true;
    return C3SC_OK;
}

int c3sc_hip_set_control_box(c3sc_hip_ctx *c, int du, const double *lb, const double *ub, int grid, int polish)
{
    if (!c || du < 1 || du > C3SC_MAX_DU || !lb || !ub || grid < 2 || polish < 0) return fail(c, C3SC_ERR_ARG, "set_control_box: bad arguments");
    double tot = 1.0;
    for (int i = 0; i < du; i++) {
        if (!(lb[i] <= ub[i])) return fail(c, C3SC_ERR_ARG, "set_control_box: lb > ub");
        c->box_lb[i] = lb[i];
        c->box_ub[i] = ub[i];
        tot *= grid;
    }
    if (tot > 1e6) return fail(c, C3SC_ERR_ARG, "set_control_box: grid^du too large");
    c->box_du = du; c->box_grid = grid; c->box_polish = polish;
    if (c->ncand == 0) { // the arena layout wants at least one candidate row; it is not read in box mode
        c->ncand = 1; c->du = du;
        c->cands.assign(du, 0.0);
        c->static_dirty = true;
    }
    return C3SC_OK;
}

int c3sc_hip_set_variant(c3sc_hip_ctx *c, int variant)
{
    if (!c) return C3SC_ERR_ARG;
    c->variant = variant;
    return C3SC_OK;
}

static int prepare_value(c3sc_hip_ctx *c, const size_t *ranks, size_t *cores_doubles)
{
    if (!c || c->d == 0 || !ranks) return fail(c, C3SC_ERR_ARG, "upload_value: set_grid first");
    const int d = c->d;
    if (ranks[0] != 1 || ranks[d] != 1) return fail(c, C3SC_ERR_ARG, "upload_value: ranks[0] and ranks[d] must be 1");
    size_t maxrank = 1;
    for (int m = 0; m <= d; m++) {
        if (ranks[m] < 1) return fail(c, C3SC_ERR_ARG, "upload_value: rank < 1");
        maxrank = std::max(maxrank, ranks[m]);
    }
    const int rp = pick_rp(d, (int)maxrank, c->model, c->variant);
    if (rp == 0) return fail(c, C3SC_ERR_UNSUPPORTED, "upload_value: no kernel instantiation for this (dim, rank)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t stat = static_layout(c);
    size_t off = stat;
    long core_off[MAXD];
    for (int m = 0; m < d; m++) {
        core_off[m] = (long)off;
        const size_t per = (m == 0 || m == d - 1) ? rp : (size_t)rp * rp;
        off += ((size_t)c->ngrid[m] * per + 15) & ~(size_t)15;
    }
    const size_t primary_end = off;
    long coreT_off[MAXD] = {0}, aop_off[MAXD] = {0};
    bool have_quad = false; // the derived copies are made only when a fiber-quad kernel of this (model, dimension, padded rank) exists
    for (const auto &e : kernel_registry())
        if (e.variant == C3SC_VARIANT_FIBER_QUAD && e.d == d && e.rp == rp && (c->model == 0 || e.model == c->model)) have_quad = true;
    if (rp % 4 == 0 && have_quad) { // derived copies for the fiber-quad kernel (k_quad_aux)
        const int C = rp / 4, MB = (C + 3) / 4;
        for (int m = 1; m < d - 1; m++) {
            coreT_off[m] = (long)off;
            off += ((size_t)c->ngrid[m] * rp * rp + 15) & ~(size_t)15;
            aop_off[m] = (long)off;
            off += ((size_t)c->ngrid[m] * 2 * MB * C * 64 + 15) & ~(size_t)15;
        }
    }
    long img_off[MAXD] = {0};
    for (int m = 0; m < d; m++) { // LDS images for the fiber-pair kernel's LDS-DMA staging (+2: its last 16-byte piece may overhang)
        const size_t per = (m == 0 || m == d - 1) ? rp : (size_t)rp * rp;
        img_off[m] = (long)off;
        off += ((size_t)c->ngrid[m] * (per | 1) + 2 + 15) & ~(size_t)15;
    }
    long qimgL_off[MAXD] = {0}, qimgR_off[MAXD] = {0};
    if (rp % 4 == 0 && have_quad) { // LDS images for the duo kernel's double-buffered LDS-DMA staging
        for (int m = 0; m < d; m++) {
            const size_t per = (m == 0 || m == d - 1) ? rp : (size_t)rp * rp;
            const size_t sz = ((size_t)c->ngrid[m] * (per + 2) + 15) & ~(size_t)15;
            if (m < d - 1) { qimgL_off[m] = (long)off; off += sz; }
            if (m > 0) { qimgR_off[m] = (long)off; off += sz; }
        }
    }
    if (off > c->arena_cap) {
        if (c->arena) HIPCHK(c, hipFree(c->arena));
        c->arena = nullptr;
        c->arena_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->arena, off * sizeof(double)));
        c->arena_cap = off;
        c->static_dirty = true;
    }
    if (stat != c->static_doubles) c->static_dirty = true;
    c->static_doubles = stat;
    for (int m = 0; m < d; m++) { c->core_off[m] = core_off[m]; c->coreT_off[m] = coreT_off[m]; c->aop_off[m] = aop_off[m]; c->img_off[m] = img_off[m]; c->qimgL_off[m] = qimgL_off[m]; c->qimgR_off[m] = qimgR_off[m]; }
    for (int m = 0; m <= d; m++) c->ranks[m] = ranks[m];
    c->rp = rp;
    *cores_doubles = primary_end - stat;
    if (c->static_dirty) return upload_static(c);
    return C3SC_OK;
}

static int make_quad_aux(c3sc_hip_ctx *c, void *stream)
{ // after the padded cores are in the arena (ordered on `stream`)
    for (int m = 1; m < c->d - 1; m++) {
        if (c->aop_off[m] == 0) continue;
        const long total = (long)c->ngrid[m] * (c->rp * c->rp + 2 * ((c->rp / 4 + 3) / 4) * (c->rp / 4) * 64);
        const int grid = (int)std::min<long>((total + 255) / 256, 1024);
        hipLaunchKernelGGL(k_quad_aux, dim3(grid), dim3(256), 0, (hipStream_t)stream, c->arena + c->core_off[m],
                           c->arena + c->coreT_off[m], c->arena + c->aop_off[m], c->ngrid[m], c->rp);
    }
    ImgJobs J; // after k_quad_aux: the suffix-side images copy the transposed cores
    J.n = 0;
    long maxtotal = 1;
    auto add = [&](long src, long dst, int nodes, int per, int stride) {
        J.src[J.n] = src; J.dst[J.n] = dst; J.nodes[J.n] = nodes; J.per[J.n] = per; J.stride[J.n] = stride;
        J.n++;
        maxtotal = std::max(maxtotal, (long)nodes * stride);
    };
    for (int m = 0; m < c->d; m++) {
        const int per = (m == 0 || m == c->d - 1) ? c->rp : c->rp * c->rp;
        add(c->core_off[m], c->img_off[m], c->ngrid[m], per, per | 1);
        if (c->qimgL_off[m]) add(c->core_off[m], c->qimgL_off[m], c->ngrid[m], per, per + 2);
        if (c->qimgR_off[m]) add((m == c->d - 1) ? c->core_off[m] : c->coreT_off[m], c->qimgR_off[m], c->ngrid[m], per, per + 2);
    }
    hipLaunchKernelGGL(k_core_images, dim3((unsigned)std::min<long>((maxtotal + 255) / 256, 64), (unsigned)J.n), dim3(256), 0, (hipStream_t)stream,
                       c->arena, J);
    HIPCHK(c, hipGetLastError());
    return C3SC_OK;
}

int c3sc_hip_upload_value(c3sc_hip_ctx *c, const size_t *ranks, const double *const *cores)
{
    size_t cd = 0;
    int rc = prepare_value(c, ranks, &cd);
    if (rc != C3SC_OK) return rc;
    const int d = c->d, rp = c->rp;
    std::vector<double> buf(cd, 0.0);
    for (int m = 0; m < d; m++) {
        const size_t r0 = ranks[m], r1 = ranks[m + 1];
        double *dst = buf.data() + (c->core_off[m] - (long)c->static_doubles);
        const double *src = cores[m];
        const size_t per = (m == 0 || m == d - 1) ? rp : (size_t)rp * rp;
        for (int j = 0; j < c->ngrid[m]; j++)
            for (size_t b = 0; b < r1; b++)
                for (size_t a = 0; a < r0; a++) {
                    const double v = src[(size_t)j * r0 * r1 + a + b * r0];
                    size_t w;
                    if (m == 0) w = b;
                    else if (m == d - 1) w = a;
                    else w = a + b * rp;
                    dst[(size_t)j * per + w] = v;
                }
    }
    HIPCHK(c, hipMemcpy(c->arena + c->static_doubles, buf.data(), cd * sizeof(double), hipMemcpyHostToDevice));
    rc = make_quad_aux(c, nullptr);
    if (rc != C3SC_OK) return rc;
    // the synchronous entry point is complete on return: the derived images are built by kernels on the NULL stream, and a caller
    // may launch on a non-blocking stream next
    HIPCHK(c, hipStreamSynchronize(nullptr));
    c->have_value = true;
    return C3SC_OK;
}

int c3sc_hip_upload_value_device(c3sc_hip_ctx *c, const size_t *ranks, const double *const *d_cores, void *stream)
{
    size_t cd = 0;
    int rc = prepare_value(c, ranks, &cd);
    if (rc != C3SC_OK) return rc;
    const int d = c->d, rp = c->rp;
    for (int m = 0; m < d; m++) {
        const int kind = (m == 0) ? 0 : (m == d - 1 ? 2 : 1);
        const long total = (long)c->ngrid[m] * (kind == 1 ? rp * rp : rp);
        const int grid = (int)std::min<long>((total + 255) / 256, 1024);
        hipLaunchKernelGGL(k_pad_core, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_cores[m], c->arena + c->core_off[m],
                           c->ngrid[m], (int)ranks[m], (int)ranks[m + 1], rp, kind);
    }
    HIPCHK(c, hipGetLastError());
    rc = make_quad_aux(c, stream);
    if (rc != C3SC_OK) return rc;
    c->have_value = true;
    return C3SC_OK;
}

static int fill_args(c3sc_hip_ctx *c, int k, size_t F, KArgs &A, bool need_model)
{
    if (!c) return C3SC_ERR_ARG;
    if (c->d == 0 || !c->have_boundary || !c->have_value) return fail(c, C3SC_ERR_ARG, "launch: grid, boundary and value must be set");
    if (need_model && (!c->have_mca || c->model == 0 || c->ncand == 0))
        return fail(c, C3SC_ERR_ARG, "launch: mca, model and controls must be set");
    if (k < 0 || k >= c->d) return fail(c, C3SC_ERR_ARG, "launch: dim_vary out of range");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->static_dirty) {
        if (static_layout(c) != c->static_doubles)
            return fail(c, C3SC_ERR_ARG, "launch: static data changed size after upload_value; upload the value again");
        int rc = upload_static(c);
        if (rc != C3SC_OK) return rc;
    }
    std::memset(&A, 0, sizeof(A));
    A.d = c->d;
    A.k = k;
    A.N = c->ngrid[k];
    A.ncand = c->ncand;
    A.F = (long)F;
    for (int m = 0; m < c->d; m++) {
        A.ngrid[m] = c->ngrid[m];
        A.bctype[m] = c->bctype[m];
        A.xg_off[m] = c->xg_off_rel[m];
        A.core_off[m] = c->core_off[m];
        A.quad_coreT_off[m] = c->coreT_off[m];
        A.quad_aop_off[m] = c->aop_off[m];
        A.pair_img_off[m] = c->img_off[m];
        A.quad_imgL_off[m] = c->qimgL_off[m];
        A.quad_imgR_off[m] = c->qimgR_off[m];
    }
    A.img_base = c->arena;
    A.cends = c->cends;
    A.memo_keys = nullptr; // only the launch paths that found a fiber-per-wave kernel switch the memo epilogue on
    A.skip = c->skip_flag;
    A.nobs = c->nobs;
    A.obs_off = c->obs_off;
    A.cands_off = c->cands_off;
    for (int t = 0; t < 4; t++) A.tab_off[t] = c->tab_off[t];
    A.cfeat_off = c->cfeat_off;
    A.h2 = c->h2;
    A.discount = c->discount;
    for (int i = 0; i < 2 * c->d; i++) A.t[i] = c->t[i];
    for (int i = 0; i < C3SC_MAX_PARAMS; i++) A.prm[i] = c->prm[i];
    A.status = c->d_status;
    { // ablation switches of the diagnostic build (make STAMPS=1); read once
        static const int dbg_env = [] { const char *e = getenv("C3SC_DBG"); return e ? atoi(e) : 0; }();
        A.dbg = dbg_env;
        if (dbg_env && !c->d_dbg) { HIPCHK(c, hipMalloc((void **)&c->d_dbg, 65536 * 8 * sizeof(unsigned long long))); } // stamp buffer
        A.dbgbuf = c->d_dbg;
        if (A.dbg & 8) A.ncand = 1;
        if (A.dbg & 16) A.ncand = 3;
    }
    return C3SC_OK;
}

// switch the memo epilogue of the fiber-per-wave kernel on for this launch if one was requested (c3sc_hip_cross_iteration)
static void arm_memo(c3sc_hip_ctx *c, const KernelEntry *e, KArgs &A)
{
    A.memo_keys = nullptr;
    if (c->memo.keys == nullptr || e->variant != C3SC_VARIANT_FIBER_PER_WAVE) return;
    A.memo_keys = c->memo.keys; A.memo_vals = c->memo.vals; A.memo_capmask = c->memo.capmask; A.memo_epoch_bits = c->memo.epoch_bits;
    A.memo_shift = c->memo.shift; A.memo_counters = c->memo.counters; A.memo_mode = c->memo.mode;
    for (int m = 0; m < c->d; m++) A.memo_stride[m] = c->memo.stride[m];
    c->memo.applied = true;
}

static int launch_bellman(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, const int32_t *d_policy, double *d_out,
                          int32_t *d_uidx, int32_t *d_absorbed, void *stream)
{
    KArgs A;
    int rc = fill_args(c, k, F, A, true);
    if (rc != C3SC_OK) return rc;
    if (F == 0) return C3SC_OK;
    if (!d_idx || !d_out) return fail(c, C3SC_ERR_ARG, "bellman_fibers: null buffer");
    A.forced = d_policy;
    LaunchIO io{c->arena, d_idx, d_out, d_uidx, d_absorbed, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream};
    // A launcher declines (nothing launched) when its LDS layout does not fit this grid or it does not serve this call; the
    // next-best instantiation of the same padded rank is tried then (e.g. the duo kernel's two staging buffers hold N <= 25 at
    // rank 16, the one-buffer quad kernel behind it N <= 75, the per-wave kernel behind that only stages the varying core).
    std::vector<const KernelEntry *> declined;
    hipError_t he = hipSuccess;
    for (;;) {
        const KernelEntry *e = find_kernel(c->model, c->d, c->rp, A.N, c->variant, k, F, &declined);
        if (!e || e->rp != c->rp) {
            if (declined.empty()) return fail(c, C3SC_ERR_UNSUPPORTED, "bellman_fibers: no kernel instantiation for (model, dim, rank, N)");
            return fail(c, C3SC_ERR_UNSUPPORTED, he == hipErrorOutOfMemory
                            ? "bellman_fibers: N x rank^2 of the varying core exceeds the 160 KB of LDS the per-wave kernel stages it in"
                            : "bellman_fibers: no kernel instantiation serves this call (model, dim, rank, N, control mode)");
        }
        c->last_kernel = e->name;
        arm_memo(c, e, A);
        he = e->fn(A, io);
        if (he != hipErrorOutOfMemory && he != hipErrorNotSupported) break;
        c->memo.applied = false;
        if (getenv("C3SC_VERBOSE")) fprintf(stderr, "c3sc: %s declined (%s), trying the next instantiation\n", e->name, hipGetErrorName(he));
        declined.push_back(e);
    }
    g_launches++;
    c->status_cache_valid = false;
    HIPCHK(c, he);
    return C3SC_OK;
}

int c3sc_hip_bellman_fibers(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, double *d_out, int32_t *d_uidx,
                            int32_t *d_absorbed, void *stream)
{
    return launch_bellman(c, k, F, d_idx, nullptr, d_out, d_uidx, d_absorbed, stream);
}

// All varying dimensions of one batch.  The launches are independent, so they are spread over the caller's stream and two side
// streams (forked from and joined to the caller's stream by events): the next dimension's workgroups fill the slots the previous
// dimension's last tiles leave.  Stream-ordered like a single launch: work enqueued on `stream` before / after the call runs
// before / after all of it.  With a memo or skip flag armed (the cross iterations' sequential steps), small AUTO batches or
// C3SC_NO_OVERLAP=1 the launches simply run in order on `stream`.
static int launch_bellman_all(c3sc_hip_ctx *c, int nk, const int *ks, const size_t *F, const int32_t *const *d_idx,
                              const int32_t *const *d_policy, double *const *d_out, int32_t *const *d_uidx, int32_t *const *d_absorbed,
                              void *stream)
{
    if (!c) return C3SC_ERR_ARG;
    if (nk < 1 || nk > MAXD || !ks || !F || !d_idx || !d_out) return fail(c, C3SC_ERR_ARG, "bellman_fibers_all: bad arguments");
    static const bool no_overlap = getenv("C3SC_NO_OVERLAP") != nullptr;
    bool overlap = nk > 1 && !no_overlap && c->memo.keys == nullptr && c->skip_flag == nullptr;
    for (int s = 0; s < nk; s++) {
        if (ks[s] < 0 || ks[s] >= c->d) return fail(c, C3SC_ERR_ARG, "bellman_fibers_all: dim_vary out of range");
        if (F[s] > 0 && (!d_idx[s] || !d_out[s])) return fail(c, C3SC_ERR_ARG, "bellman_fibers_all: null buffer");
        if (d_policy && !d_policy[s]) return fail(c, C3SC_ERR_ARG, "bellman_fibers_all: null policy");
        for (int q = 0; q < s; q++)
            if (d_out[q] == d_out[s] && F[s] > 0) return fail(c, C3SC_ERR_ARG, "bellman_fibers_all: two segments share an output array");
    }
    hipStream_t st = (hipStream_t)stream;
    if (overlap) {
        HIPCHK(c, hipSetDevice(c->device));
        if (!c->fork_ev) {
            HIPCHK(c, hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
            for (int i = 0; i < c3sc_hip_ctx::NSIDE; i++) {
                HIPCHK(c, hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
                HIPCHK(c, hipEventCreateWithFlags(&c->join_ev[i], hipEventDisableTiming));
            }
        }
        HIPCHK(c, hipEventRecord(c->fork_ev, st));
        for (int i = 0; i < c3sc_hip_ctx::NSIDE; i++) HIPCHK(c, hipStreamWaitEvent(c->side[i], c->fork_ev, 0));
    }
    int rc = C3SC_OK;
    for (int s = 0; s < nk && rc == C3SC_OK; s++) {
        const int lane = overlap ? s % (c3sc_hip_ctx::NSIDE + 1) : 0; // 0: the caller's stream
        rc = launch_bellman(c, ks[s], F[s], d_idx[s], d_policy ? d_policy[s] : nullptr, d_out[s], d_uidx ? d_uidx[s] : nullptr,
                            d_absorbed ? d_absorbed[s] : nullptr, lane == 0 ? stream : (void *)c->side[lane - 1]);
    }
    if (overlap) // join even after a failed launch: what was enqueued on the side streams must not outlive the call's ordering
        for (int i = 0; i < c3sc_hip_ctx::NSIDE; i++) {
            HIPCHK(c, hipEventRecord(c->join_ev[i], c->side[i]));
            HIPCHK(c, hipStreamWaitEvent(st, c->join_ev[i], 0));
        }
    return rc;
}

int c3sc_hip_bellman_fibers_all(c3sc_hip_ctx *c, int nk, const int *ks, const size_t *F, const int32_t *const *d_idx, double *const *d_out,
                                int32_t *const *d_uidx, int32_t *const *d_absorbed, void *stream)
{
    return launch_bellman_all(c, nk, ks, F, d_idx, nullptr, d_out, d_uidx, d_absorbed, stream);
}

int c3sc_hip_policy_fibers_all(c3sc_hip_ctx *c, int nk, const int *ks, const size_t *F, const int32_t *const *d_idx,
                               const int32_t *const *d_policy, double *const *d_out, int32_t *const *d_absorbed, void *stream)
{
    if (!d_policy) return fail(c, C3SC_ERR_ARG, "policy_fibers_all: null policy");
    return launch_bellman_all(c, nk, ks, F, d_idx, d_policy, d_out, nullptr, d_absorbed, stream);
}

static int launch_box(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, const double *d_policy_u, double *d_out,
                      double *d_uopt, int32_t *d_absorbed, void *stream)
{
    if (!c) return C3SC_ERR_ARG;
    if (c->box_du == 0) return fail(c, C3SC_ERR_ARG, "bellman_fibers_box: c3sc_hip_set_control_box first");
    if (model_ncf(c->model) != 0 && c->model != C3SC_MODEL_COTHRUST6D) // cothrust forms its features from u on the device (models.hpp: CF_FROM_U)
        return fail(c, C3SC_ERR_UNSUPPORTED, "bellman_fibers_box: this model needs transcendental functions of the control");
    KArgs A;
    int rc = fill_args(c, k, F, A, true);
    if (rc != C3SC_OK) return rc;
    if (F == 0) return C3SC_OK;
    if (!d_idx || !d_out) return fail(c, C3SC_ERR_ARG, "bellman_fibers_box: null buffer");
    A.cmode = 1;
    A.ugrid = c->box_grid;
    A.upolish = c->box_polish;
    for (int i = 0; i < c->box_du; i++) { A.ulb[i] = c->box_lb[i]; A.uub[i] = c->box_ub[i]; }
    A.uopt = d_uopt;
    A.forced_u = d_policy_u;
    const KernelEntry *e = find_kernel(c->model, c->d, c->rp, A.N, C3SC_VARIANT_FIBER_PER_WAVE, k);
    if (!e || e->rp != c->rp) return fail(c, C3SC_ERR_UNSUPPORTED, "bellman_fibers_box: no fiber-per-wave instantiation for (model, dim, rank, N)");
    c->last_kernel = e->name;
    arm_memo(c, e, A);
    LaunchIO io{c->arena, d_idx, d_out, nullptr, d_absorbed, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream};
    g_launches++;
    c->status_cache_valid = false;
    const hipError_t he = e->fn(A, io);
    if (he == hipErrorNotSupported) return fail(c, C3SC_ERR_UNSUPPORTED, "bellman_fibers_box: no box-minimiser instantiation for this model");
    HIPCHK(c, he);
    return C3SC_OK;
}

int c3sc_hip_bellman_fibers_box(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, double *d_out, double *d_uopt,
                                int32_t *d_absorbed, void *stream)
{
    return launch_box(c, k, F, d_idx, nullptr, d_out, d_uopt, d_absorbed, stream);
}

int c3sc_hip_policy_fibers_box(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, const double *d_policy_u, double *d_out,
                               int32_t *d_absorbed, void *stream)
{
    if (F != 0 && !d_policy_u) return fail(c, C3SC_ERR_ARG, "policy_fibers_box: null policy");
    return launch_box(c, k, F, d_idx, d_policy_u, d_out, nullptr, d_absorbed, stream);
}

static int box_host(c3sc_hip_ctx *c, int k, size_t F, const int32_t *h_idx, const double *h_policy_u, double *h_out, double *h_uopt,
                    int32_t *h_absorbed)
{
    if (!c || c->d == 0 || k < 0 || k >= c->d || c->box_du == 0) return fail(c, C3SC_ERR_ARG, "bellman_fibers_box_host: bad arguments");
    if (F == 0) return C3SC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = c->ngrid[k], du = c->box_du;
    const size_t b_idx = align256(F * c->d * sizeof(int32_t)), b_out = align256(F * N * sizeof(double)),
                 b_u = align256(F * N * du * sizeof(double)), b_i = align256(F * N * sizeof(int32_t));
    int rc;
    if (zero_copy_batch(b_idx + b_out + b_u + b_i)) {
        if ((rc = ensure_pinned(c, b_idx + b_out + b_u + b_i)) != C3SC_OK) return rc;
        char *hb = (char *)c->pinned, *db = (char *)c->pinned_dev;
        memcpy(hb, h_idx, F * c->d * sizeof(int32_t));
        double *zu = (double *)(db + b_idx + b_out);
        int32_t *zab = h_absorbed ? (int32_t *)(db + b_idx + b_out + b_u) : nullptr;
        if (h_policy_u) {
            memcpy(hb + b_idx + b_out, h_policy_u, F * N * du * sizeof(double));
            rc = c3sc_hip_policy_fibers_box(c, k, F, (int32_t *)db, zu, (double *)(db + b_idx), zab, nullptr);
        } else {
            rc = c3sc_hip_bellman_fibers_box(c, k, F, (int32_t *)db, (double *)(db + b_idx), h_uopt ? zu : nullptr, zab, nullptr);
        }
        if (rc != C3SC_OK) return rc;
        HIPCHK(c, hipStreamSynchronize(nullptr));
        memcpy(h_out, hb + b_idx, F * N * sizeof(double));
        if (h_uopt && !h_policy_u) memcpy(h_uopt, hb + b_idx + b_out, F * N * du * sizeof(double));
        if (h_absorbed) memcpy(h_absorbed, hb + b_idx + b_out + b_u, F * N * sizeof(int32_t));
        return C3SC_OK;
    }
    rc = ensure_scratch(c, b_idx + b_out + b_u + b_i);
    if (rc != C3SC_OK) return rc;
    char *base = (char *)c->scratch;
    int32_t *d_idx = (int32_t *)base;
    double *d_out = (double *)(base + b_idx);
    double *d_u = (double *)(base + b_idx + b_out);
    int32_t *d_ab = (int32_t *)(base + b_idx + b_out + b_u);
    HIPCHK(c, hipMemcpy(d_idx, h_idx, F * c->d * sizeof(int32_t), hipMemcpyHostToDevice));
    if (h_policy_u) {
        HIPCHK(c, hipMemcpy(d_u, h_policy_u, F * N * du * sizeof(double), hipMemcpyHostToDevice));
        rc = c3sc_hip_policy_fibers_box(c, k, F, d_idx, d_u, d_out, h_absorbed ? d_ab : nullptr, nullptr);
    } else {
        rc = c3sc_hip_bellman_fibers_box(c, k, F, d_idx, d_out, h_uopt ? d_u : nullptr, h_absorbed ? d_ab : nullptr, nullptr);
    }
    if (rc != C3SC_OK) return rc;
    HIPCHK(c, hipMemcpy(h_out, d_out, F * N * sizeof(double), hipMemcpyDeviceToHost));
    if (h_uopt && !h_policy_u) HIPCHK(c, hipMemcpy(h_uopt, d_u, F * N * du * sizeof(double), hipMemcpyDeviceToHost));
    if (h_absorbed) HIPCHK(c, hipMemcpy(h_absorbed, d_ab, F * N * sizeof(int32_t), hipMemcpyDeviceToHost));
    return C3SC_OK;
}

int c3sc_hip_bellman_fibers_box_host(c3sc_hip_ctx *c, int k, size_t F, const int32_t *h_idx, double *h_out, double *h_uopt,
                                     int32_t *h_absorbed)
{
    return box_host(c, k, F, h_idx, nullptr, h_out, h_uopt, h_absorbed);
}

int c3sc_hip_policy_fibers_box_host(c3sc_hip_ctx *c, int k, size_t F, const int32_t *h_idx, const double *h_policy_u, double *h_out,
                                    int32_t *h_absorbed)
{
    if (F != 0 && !h_policy_u) return fail(c, C3SC_ERR_ARG, "policy_fibers_box_host: null policy");
    return box_host(c, k, F, h_idx, h_policy_u, h_out, nullptr, h_absorbed);
}

int c3sc_hip_policy_fibers(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, const int32_t *d_policy, double *d_out,
                           int32_t *d_absorbed, void *stream)
{
    if (F != 0 && !d_policy) return fail(c, C3SC_ERR_ARG, "policy_fibers: null policy");
    return launch_bellman(c, k, F, d_idx, d_policy, d_out, nullptr, d_absorbed, stream);
}

static int launch_tables(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, const double *d_tables, const double *d_costs2,
                         const int32_t *d_policy, double *d_out, int32_t *d_uidx, int32_t *d_absorbed, void *stream)
{
    if (!c) return C3SC_ERR_ARG;
    const int saved_model = c->model;
    c->model = C3SC_MODEL_TABLE; // fill_args only checks that a model is set
    KArgs A;
    int rc = fill_args(c, k, F, A, true);
    c->model = saved_model;
    if (rc != C3SC_OK) return rc;
    if (F == 0) return C3SC_OK;
    if (!d_idx || !d_out || !d_tables || !d_costs2) return fail(c, C3SC_ERR_ARG, "bellman_fibers_tables: null buffer");
    A.forced = d_policy;
    const KernelEntry *e = find_kernel(C3SC_MODEL_TABLE, c->d, c->rp, A.N, C3SC_VARIANT_AUTO, k);
    if (!e || e->rp != c->rp) return fail(c, C3SC_ERR_UNSUPPORTED, "bellman_fibers_tables: no kernel instantiation for (dim, rank, N)");
    c->last_kernel = e->name;
    LaunchIO io{c->arena, d_idx, d_out, d_uidx, d_absorbed, nullptr, nullptr, d_tables, d_costs2, (hipStream_t)stream};
    g_launches++;
    c->status_cache_valid = false;
    HIPCHK(c, e->fn(A, io));
    return C3SC_OK;
}

int c3sc_hip_bellman_fibers_tables(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, const double *d_tables,
                                   const double *d_costs2, double *d_out, int32_t *d_uidx, int32_t *d_absorbed, void *stream)
{
    return launch_tables(c, k, F, d_idx, d_tables, d_costs2, nullptr, d_out, d_uidx, d_absorbed, stream);
}

int c3sc_hip_policy_fibers_tables(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, const double *d_tables,
                                  const double *d_costs2, const int32_t *d_policy, double *d_out, int32_t *d_absorbed, void *stream)
{
    if (F != 0 && !d_policy) return fail(c, C3SC_ERR_ARG, "policy_fibers_tables: null policy");
    return launch_tables(c, k, F, d_idx, d_tables, d_costs2, d_policy, d_out, nullptr, d_absorbed, stream);
}

static int tables_host(c3sc_hip_ctx *c, int k, size_t F, const int32_t *h_idx, const double *h_tables, const double *h_costs2,
                       const int32_t *h_policy, double *h_out, int32_t *h_uidx, int32_t *h_absorbed)
{
    if (!c || c->d == 0 || k < 0 || k >= c->d || c->ncand == 0) return fail(c, C3SC_ERR_ARG, "bellman_fibers_tables_host: bad arguments");
    if (F == 0) return C3SC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = c->ngrid[k], S = 2 * c->d + 1;
    const size_t b_idx = align256(F * c->d * sizeof(int32_t)), b_out = align256(F * N * sizeof(double)),
                 b_i = align256(F * N * sizeof(int32_t)), b_t = align256(F * N * c->ncand * S * sizeof(double)),
                 b_c = align256(F * N * 2 * sizeof(double));
    int rc = ensure_scratch(c, b_idx + b_out + 2 * b_i + b_t + b_c);
    if (rc != C3SC_OK) return rc;
    char *base = (char *)c->scratch;
    int32_t *d_idx = (int32_t *)base;
    double *d_out = (double *)(base + b_idx);
    int32_t *d_ui = (int32_t *)(base + b_idx + b_out);
    int32_t *d_ab = (int32_t *)(base + b_idx + b_out + b_i);
    double *d_t = (double *)(base + b_idx + b_out + 2 * b_i);
    double *d_c = (double *)(base + b_idx + b_out + 2 * b_i + b_t);
    HIPCHK(c, hipMemcpy(d_idx, h_idx, F * c->d * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_t, h_tables, F * N * c->ncand * S * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_c, h_costs2, F * N * 2 * sizeof(double), hipMemcpyHostToDevice));
    if (h_policy) { // the uidx buffer carries the policy in
        HIPCHK(c, hipMemcpy(d_ui, h_policy, F * N * sizeof(int32_t), hipMemcpyHostToDevice));
        rc = c3sc_hip_policy_fibers_tables(c, k, F, d_idx, d_t, d_c, d_ui, d_out, h_absorbed ? d_ab : nullptr, nullptr);
    } else {
        rc = c3sc_hip_bellman_fibers_tables(c, k, F, d_idx, d_t, d_c, d_out, h_uidx ? d_ui : nullptr, h_absorbed ? d_ab : nullptr, nullptr);
    }
    if (rc != C3SC_OK) return rc;
    HIPCHK(c, hipMemcpy(h_out, d_out, F * N * sizeof(double), hipMemcpyDeviceToHost));
    if (h_uidx && !h_policy) HIPCHK(c, hipMemcpy(h_uidx, d_ui, F * N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (h_absorbed) HIPCHK(c, hipMemcpy(h_absorbed, d_ab, F * N * sizeof(int32_t), hipMemcpyDeviceToHost));
    return C3SC_OK;
}

int c3sc_hip_bellman_fibers_tables_host(c3sc_hip_ctx *c, int k, size_t F, const int32_t *h_idx, const double *h_tables,
                                        const double *h_costs2, double *h_out, int32_t *h_uidx, int32_t *h_absorbed)
{
    return tables_host(c, k, F, h_idx, h_tables, h_costs2, nullptr, h_out, h_uidx, h_absorbed);
}

int c3sc_hip_policy_fibers_tables_host(c3sc_hip_ctx *c, int k, size_t F, const int32_t *h_idx, const double *h_tables,
                                       const double *h_costs2, const int32_t *h_policy, double *h_out, int32_t *h_absorbed)
{
    if (F != 0 && !h_policy) return fail(c, C3SC_ERR_ARG, "policy_fibers_tables_host: null policy");
    return tables_host(c, k, F, h_idx, h_tables, h_costs2, h_policy, h_out, nullptr, h_absorbed);
}

int c3sc_hip_stencil_fibers(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, double *d_costs, int32_t *d_absorbed,
                            void *stream)
{
    return c3sc_hip_stencil_fibers_nb(c, k, F, d_idx, nullptr, nullptr, d_costs, d_absorbed, stream);
}

int c3sc_hip_stencil_fibers_nb(c3sc_hip_ctx *c, int k, size_t F, const int32_t *d_idx, const int32_t *d_nb_fixed,
                               const int32_t *d_nb_vary, double *d_costs, int32_t *d_absorbed, void *stream)
{
    KArgs A;
    int rc = fill_args(c, k, F, A, false);
    if (rc != C3SC_OK) return rc;
    if (F == 0) return C3SC_OK;
    if (!d_idx || !d_costs) return fail(c, C3SC_ERR_ARG, "stencil_fibers: null buffer");
    const KernelEntry *e = find_kernel(0, c->d, c->rp, A.N, C3SC_VARIANT_AUTO, k);
    if (!e || e->rp != c->rp) return fail(c, C3SC_ERR_UNSUPPORTED, "stencil_fibers: no kernel instantiation for (dim, rank, N)");
    c->last_kernel = e->name;
    LaunchIO io{c->arena, d_idx, d_costs, nullptr, d_absorbed, d_nb_fixed, d_nb_vary, nullptr, nullptr, (hipStream_t)stream};
    g_launches++;
    c->status_cache_valid = false;
    HIPCHK(c, e->fn(A, io));
    return C3SC_OK;
}

static int ensure_scratch(c3sc_hip_ctx *c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return C3SC_OK;
    if (c->scratch) HIPCHK(c, hipFree(c->scratch));
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    HIPCHK(c, hipMalloc(&c->scratch, bytes));
    c->scratch_bytes = bytes;
    return C3SC_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Small batches of the *_host entry points (a cross-approximation core step: ~100 fibers, 3 KB in, 33 KB out) skip
// the staging copies: indices are placed in a pinned host block the device maps, the kernel reads them and writes its
// rows there, and the call is launch + stream synchronise.  The two hipMemcpy calls they replace cost 25 us of a
// 53 us call (tools/call_latency.py).  Large batches keep the copies: PCIe would bound the kernel.
static constexpr size_t ZERO_COPY_MAX_BYTES = (size_t)1 << 20;
static bool zero_copy_batch(size_t bytes)
{
    static const bool off = getenv("C3SC_NO_ZEROCOPY") != nullptr;
    return !off && bytes <= ZERO_COPY_MAX_BYTES;
}
static int ensure_pinned(c3sc_hip_ctx *c, size_t bytes)
{
    if (bytes <= c->pinned_bytes) return C3SC_OK;
    if (c->pinned) HIPCHK(c, hipHostFree(c->pinned));
    c->pinned = c->pinned_dev = nullptr;
    c->pinned_bytes = 0;
    const size_t cap = bytes < ((size_t)256 << 10) ? ((size_t)256 << 10) : bytes;
    HIPCHK(c, hipHostMalloc(&c->pinned, cap, hipHostMallocMapped | hipHostMallocPortable));
    HIPCHK(c, hipHostGetDevicePointer(&c->pinned_dev, c->pinned, 0));
    c->pinned_bytes = cap;
    return C3SC_OK;
}

int c3sc_hip_bellman_fibers_host(c3sc_hip_ctx *c, int k, size_t F, const int32_t *h_idx, double *h_out, int32_t *h_uidx,
                                 int32_t *h_absorbed)
{
    if (!c || c->d == 0 || k < 0 || k >= c->d) return fail(c, C3SC_ERR_ARG, "bellman_fibers_host: bad arguments");
    if (F == 0) return C3SC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = c->ngrid[k];
    const size_t b_idx = align256(F * c->d * sizeof(int32_t)), b_out = align256(F * N * sizeof(double)),
                 b_i = align256(F * N * sizeof(int32_t));
    int rc;
    if (zero_copy_batch(b_idx + b_out + 2 * b_i)) {
        if ((rc = ensure_pinned(c, b_idx + b_out + 2 * b_i)) != C3SC_OK) return rc;
        char *hb = (char *)c->pinned, *db = (char *)c->pinned_dev;
        memcpy(hb, h_idx, F * c->d * sizeof(int32_t));
        rc = c3sc_hip_bellman_fibers(c, k, F, (int32_t *)db, (double *)(db + b_idx), h_uidx ? (int32_t *)(db + b_idx + b_out) : nullptr,
                                     h_absorbed ? (int32_t *)(db + b_idx + b_out + b_i) : nullptr, nullptr);
        if (rc != C3SC_OK) return rc;
        HIPCHK(c, hipStreamSynchronize(nullptr));
        memcpy(h_out, hb + b_idx, F * N * sizeof(double));
        if (h_uidx) memcpy(h_uidx, hb + b_idx + b_out, F * N * sizeof(int32_t));
        if (h_absorbed) memcpy(h_absorbed, hb + b_idx + b_out + b_i, F * N * sizeof(int32_t));
        return C3SC_OK;
    }
    rc = ensure_scratch(c, b_idx + b_out + 2 * b_i);
    if (rc != C3SC_OK) return rc;
    char *base = (char *)c->scratch;
    int32_t *d_idx = (int32_t *)base;
    double *d_out = (double *)(base + b_idx);
    int32_t *d_ui = (int32_t *)(base + b_idx + b_out);
    int32_t *d_ab = (int32_t *)(base + b_idx + b_out + b_i);
    HIPCHK(c, hipMemcpy(d_idx, h_idx, F * c->d * sizeof(int32_t), hipMemcpyHostToDevice));
    rc = c3sc_hip_bellman_fibers(c, k, F, d_idx, d_out, h_uidx ? d_ui : nullptr, h_absorbed ? d_ab : nullptr, nullptr);
    if (rc != C3SC_OK) return rc;
    HIPCHK(c, hipMemcpy(h_out, d_out, F * N * sizeof(double), hipMemcpyDeviceToHost));
    if (h_uidx) HIPCHK(c, hipMemcpy(h_uidx, d_ui, F * N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (h_absorbed) HIPCHK(c, hipMemcpy(h_absorbed, d_ab, F * N * sizeof(int32_t), hipMemcpyDeviceToHost));
    return C3SC_OK;
}

int c3sc_hip_policy_fibers_host(c3sc_hip_ctx *c, int k, size_t F, const int32_t *h_idx, const int32_t *h_policy, double *h_out,
                                int32_t *h_absorbed)
{
    if (!c || c->d == 0 || k < 0 || k >= c->d || (F != 0 && !h_policy)) return fail(c, C3SC_ERR_ARG, "policy_fibers_host: bad arguments");
    if (F == 0) return C3SC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = c->ngrid[k];
    const size_t b_idx = align256(F * c->d * sizeof(int32_t)), b_out = align256(F * N * sizeof(double)),
                 b_i = align256(F * N * sizeof(int32_t));
    int rc;
    if (zero_copy_batch(b_idx + b_out + 2 * b_i)) {
        if ((rc = ensure_pinned(c, b_idx + b_out + 2 * b_i)) != C3SC_OK) return rc;
        char *hb = (char *)c->pinned, *db = (char *)c->pinned_dev;
        memcpy(hb, h_idx, F * c->d * sizeof(int32_t));
        memcpy(hb + b_idx + b_out, h_policy, F * N * sizeof(int32_t));
        rc = c3sc_hip_policy_fibers(c, k, F, (int32_t *)db, (int32_t *)(db + b_idx + b_out), (double *)(db + b_idx),
                                    h_absorbed ? (int32_t *)(db + b_idx + b_out + b_i) : nullptr, nullptr);
        if (rc != C3SC_OK) return rc;
        HIPCHK(c, hipStreamSynchronize(nullptr));
        memcpy(h_out, hb + b_idx, F * N * sizeof(double));
        if (h_absorbed) memcpy(h_absorbed, hb + b_idx + b_out + b_i, F * N * sizeof(int32_t));
        return C3SC_OK;
    }
    rc = ensure_scratch(c, b_idx + b_out + 2 * b_i);
    if (rc != C3SC_OK) return rc;
    char *base = (char *)c->scratch;
    int32_t *d_idx = (int32_t *)base;
    double *d_out = (double *)(base + b_idx);
    int32_t *d_pol = (int32_t *)(base + b_idx + b_out);
    int32_t *d_ab = (int32_t *)(base + b_idx + b_out + b_i);
    HIPCHK(c, hipMemcpy(d_idx, h_idx, F * c->d * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_pol, h_policy, F * N * sizeof(int32_t), hipMemcpyHostToDevice));
    rc = c3sc_hip_policy_fibers(c, k, F, d_idx, d_pol, d_out, h_absorbed ? d_ab : nullptr, nullptr);
    if (rc != C3SC_OK) return rc;
    HIPCHK(c, hipMemcpy(h_out, d_out, F * N * sizeof(double), hipMemcpyDeviceToHost));
    if (h_absorbed) HIPCHK(c, hipMemcpy(h_absorbed, d_ab, F * N * sizeof(int32_t), hipMemcpyDeviceToHost));
    return C3SC_OK;
}

int c3sc_hip_stencil_fibers_host(c3sc_hip_ctx *c, int k, size_t F, const int32_t *h_idx, double *h_costs, int32_t *h_absorbed)
{
    return c3sc_hip_stencil_fibers_nb_host(c, k, F, h_idx, nullptr, nullptr, h_costs, h_absorbed);
}

int c3sc_hip_stencil_fibers_nb_host(c3sc_hip_ctx *c, int k, size_t F, const int32_t *h_idx, const int32_t *h_nb_fixed,
                                    const int32_t *h_nb_vary, double *h_costs, int32_t *h_absorbed)
{
    if (!c || c->d == 0 || k < 0 || k >= c->d) return fail(c, C3SC_ERR_ARG, "stencil_fibers_host: bad arguments");
    if (F == 0) return C3SC_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = c->ngrid[k], S = 2 * c->d + 1;
    const size_t b_idx = align256(F * c->d * sizeof(int32_t)), b_out = align256(F * N * S * sizeof(double)),
                 b_i = align256(F * N * sizeof(int32_t)), b_nf = align256(F * 2 * (c->d - 1) * sizeof(int32_t)),
                 b_nv = align256(F * N * 2 * sizeof(int32_t));
    int rc = ensure_scratch(c, b_idx + b_out + b_i + b_nf + b_nv);
    if (rc != C3SC_OK) return rc;
    char *base = (char *)c->scratch;
    int32_t *d_idx = (int32_t *)base;
    double *d_out = (double *)(base + b_idx);
    int32_t *d_ab = (int32_t *)(base + b_idx + b_out);
    int32_t *d_nf = (int32_t *)(base + b_idx + b_out + b_i);
    int32_t *d_nv = (int32_t *)(base + b_idx + b_out + b_i + b_nf);
    HIPCHK(c, hipMemcpy(d_idx, h_idx, F * c->d * sizeof(int32_t), hipMemcpyHostToDevice));
    if (h_nb_fixed) HIPCHK(c, hipMemcpy(d_nf, h_nb_fixed, F * 2 * (c->d - 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    if (h_nb_vary) HIPCHK(c, hipMemcpy(d_nv, h_nb_vary, F * N * 2 * sizeof(int32_t), hipMemcpyHostToDevice));
    rc = c3sc_hip_stencil_fibers_nb(c, k, F, d_idx, h_nb_fixed ? d_nf : nullptr, h_nb_vary ? d_nv : nullptr, d_out,
                                    h_absorbed ? d_ab : nullptr, nullptr);
    if (rc != C3SC_OK) return rc;
    HIPCHK(c, hipMemcpy(h_costs, d_out, F * N * S * sizeof(double), hipMemcpyDeviceToHost));
    if (h_absorbed) HIPCHK(c, hipMemcpy(h_absorbed, d_ab, F * N * sizeof(int32_t), hipMemcpyDeviceToHost));
    return C3SC_OK;
}

int c3sc_hip_sync(c3sc_hip_ctx *c, void *stream)
{
    if (!c) return C3SC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
    return C3SC_OK;
}

int c3sc_hip_get_status(c3sc_hip_ctx *c, unsigned *flags, int clear)
{
    if (!c || !flags) return C3SC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->status_cache_valid) *flags = c->status_cache; /* read with the last cross fetch; nothing was launched since */
    else HIPCHK(c, hipMemcpy(flags, c->d_status, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (clear && *flags) { /* nothing to clear otherwise */
        HIPCHK(c, hipMemset(c->d_status, 0, sizeof(unsigned)));
        c->status_cache = 0;
    }
    return C3SC_OK;
}

unsigned long long c3sc_hip_launch_count(void) { return g_launches; }

int c3sc_hip_debug_read(c3sc_hip_ctx *c, unsigned long long *out, size_t n)
{
    if (!c || !c->d_dbg || !out) return C3SC_ERR_ARG;
    HIPCHK(c, hipMemcpy(out, c->d_dbg, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return C3SC_OK;
}

int c3sc_hip_timer_start(c3sc_hip_ctx *c, void *stream)
{
    if (!c) return C3SC_ERR_ARG;
    HIPCHK(c, hipEventRecord(c->ev0, (hipStream_t)stream));
    return C3SC_OK;
}

int c3sc_hip_timer_stop(c3sc_hip_ctx *c, void *stream, float *ms)
{
    if (!c || !ms) return C3SC_ERR_ARG;
    HIPCHK(c, hipEventRecord(c->ev1, (hipStream_t)stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return C3SC_OK;
}

static int run_peak(c3sc_hip_ctx *c, bool mfma, double *tflops)
{
    if (!c || !tflops) return C3SC_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const char *ev = getenv("C3SC_PEAK_BLOCKS_PER_CU");
    const int blocks = 256 * (ev ? atoi(ev) : 8), threads = 256, iters = 20000;
    int rc = ensure_scratch(c, (size_t)blocks * threads * sizeof(double));
    if (rc != C3SC_OK) return rc;
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        HIPCHK(c, hipEventRecord(c->ev0, nullptr));
        if (mfma) hipLaunchKernelGGL(k_peak_mfma, dim3(blocks), dim3(threads), 0, nullptr, (double *)c->scratch, iters);
        else hipLaunchKernelGGL(k_peak_fma, dim3(blocks), dim3(threads), 0, nullptr, (double *)c->scratch, iters);
        HIPCHK(c, hipEventRecord(c->ev1, nullptr));
        HIPCHK(c, hipEventSynchronize(c->ev1));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double waves = (double)blocks * threads / 64.0;
    const double flop = mfma ? waves * iters * 4.0 * (2.0 * 16 * 16 * 4) : (double)blocks * threads * iters * 8.0 * 2.0;
    *tflops = flop / (best * 1e-3) / 1e12;
    return C3SC_OK;
}

int c3sc_hip_peak_fma_f64(c3sc_hip_ctx *c, double *tflops) { return run_peak(c, false, tflops); }
int c3sc_hip_peak_mfma_f64(c3sc_hip_ctx *c, double *tflops) { return run_peak(c, true, tflops); }

} // extern "C"
