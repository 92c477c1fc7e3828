/* c3sc_bellman.c -- host side of the Bellman backup behind the reference's C API: nodal ValueF,
 * MCA helpers, parameter bundles, and bellman_vi, which keeps the cross-approximation callback ABI
 * (src/valuefunc.c:615-616) and runs the fiber on the MI355X through include/c3sc_hip.h.
 * Own implementation of the reference's interface; citations are relative to the reference tree. */
#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "c3sc/c3sc.h"
#include "c3sc_hip.h"
#include "c3sc_private.h"

void *c3sc_xcalloc(size_t n, size_t s)
{
    void *p = calloc(n ? n : 1, s);
    if (p == NULL) DIE("out of memory");
    return p;
}

static void hipok(struct c3sc_hip_ctx *ctx, int rc, const char *what)
{
    if (rc != C3SC_OK) DIE("%s failed (code %d): %s", what, rc, c3sc_hip_last_error(ctx));
}

/* =============================================================================== ValueF (struct in c3sc_private.h) */
static unsigned long g_vf_version = 0;
/* which value function is resident on which device context (a workspace owns at most two) */
#define MAX_TRACKED_CTX 16
static struct { struct c3sc_hip_ctx *ctx; unsigned long version; uint64_t cfg_sig; int cfg_set; } g_ctx[MAX_TRACKED_CTX];
static int ctx_slot(struct c3sc_hip_ctx *ctx)
{
    int free_slot = -1;
    for (int i = 0; i < MAX_TRACKED_CTX; i++) {
        if (g_ctx[i].ctx == ctx) return i;
        if (g_ctx[i].ctx == NULL && free_slot < 0) free_slot = i;
    }
    if (free_slot < 0) { free_slot = 0; } /* recycle: worst case is one redundant upload */
    g_ctx[free_slot].ctx = ctx;
    g_ctx[free_slot].version = 0;
    g_ctx[free_slot].cfg_set = 0;
    return free_slot;
}

struct ValueF *valuef_create_nodal(size_t d, const size_t *N, const size_t *ranks, double **cores)
{
    struct ValueF *vf = xcalloc(1, sizeof(*vf));
    vf->d = d;
    vf->N = xcalloc(d, sizeof(size_t));
    vf->ranks = xcalloc(d + 1, sizeof(size_t));
    vf->cores = xcalloc(d, sizeof(double *));
    memcpy(vf->N, N, d * sizeof(size_t));
    memcpy(vf->ranks, ranks, (d + 1) * sizeof(size_t));
    for (size_t m = 0; m < d; m++) {
        const size_t n = N[m] * ranks[m] * ranks[m + 1];
        vf->cores[m] = xcalloc(n, sizeof(double));
        memcpy(vf->cores[m], cores[m], n * sizeof(double));
    }
    vf->version = ++g_vf_version;
    return vf;
}

void valuef_attach_grid(struct ValueF *vf, double **grid)
{ /* the nodes the cores are sampled on: needed by the continuous norms and off-grid evaluation */
    if (vf->grid == NULL) vf->grid = xcalloc(vf->d, sizeof(double *));
    for (size_t m = 0; m < vf->d; m++) {
        free(vf->grid[m]);
        vf->grid[m] = xcalloc(vf->N[m], sizeof(double));
        memcpy(vf->grid[m], grid[m], vf->N[m] * sizeof(double));
    }
}

void valuef_set_cross_indices(struct ValueF *vf, const size_t *nisl, int *const *isl, const size_t *nisr, int *const *isr)
{ /* left sets: nisl[k] tuples over dims 0..k-1; right sets: nisr[k] tuples over dims k+1..d-1 */
    valuef_free_cross_indices(vf);
    const size_t d = vf->d;
    vf->nisl = xcalloc(d, sizeof(size_t)); vf->nisr = xcalloc(d, sizeof(size_t));
    vf->isl = xcalloc(d, sizeof(int *)); vf->isr = xcalloc(d, sizeof(int *));
    for (size_t k = 0; k < d; k++) {
        vf->nisl[k] = nisl[k]; vf->nisr[k] = nisr[k];
        vf->isl[k] = xcalloc(nisl[k] * (k ? k : 1), sizeof(int));
        vf->isr[k] = xcalloc(nisr[k] * (d - 1 - k ? d - 1 - k : 1), sizeof(int));
        if (k > 0) memcpy(vf->isl[k], isl[k], nisl[k] * k * sizeof(int));
        if (k + 1 < d) memcpy(vf->isr[k], isr[k], nisr[k] * (d - 1 - k) * sizeof(int));
    }
}

void valuef_free_cross_indices(struct ValueF *vf)
{
    if (vf->isl) for (size_t k = 0; k < vf->d; k++) { free(vf->isl[k]); free(vf->isr[k]); }
    free(vf->isl); free(vf->isr); free(vf->nisl); free(vf->nisr);
    vf->isl = vf->isr = NULL; vf->nisl = vf->nisr = NULL;
}

void valuef_destroy(struct ValueF *vf)
{
    if (vf == NULL) return;
    for (size_t m = 0; m < vf->d; m++) { free(vf->cores[m]); if (vf->grid) free(vf->grid[m]); }
    valuef_free_cross_indices(vf);
    free(vf->grid); free(vf->cores); free(vf->N); free(vf->ranks); free(vf);
}

struct ValueF *valuef_copy(struct ValueF *vf)
{ /* valuefunc.c:126-158: the copy keeps the cross index sets (the next interpolation warm-starts from them) */
    struct ValueF *c = valuef_create_nodal(vf->d, vf->N, vf->ranks, vf->cores);
    if (vf->grid) valuef_attach_grid(c, vf->grid);
    if (vf->isl) valuef_set_cross_indices(c, vf->nisl, vf->isl, vf->nisr, vf->isr);
    c->sets_stable = vf->sets_stable;
    c->elem_class = vf->elem_class;
    return c;
}
size_t *valuef_get_ranks(struct ValueF *vf) { return vf->ranks; }
/* valuefunc.c:218-221 hands out C3's left cross-index sets; here the handle is this library's own tuple sets
 * (isl[k]: nisl[k] tuples over dims 0..k-1, or NULL before the first interpolation) -- opaque to callers */
struct CrossIndex **valuef_get_isl(const struct ValueF *vf) { return (struct CrossIndex **)vf->isl; }
size_t valuef_get_dim(const struct ValueF *vf) { return vf->d; }
const size_t *valuef_get_N(const struct ValueF *vf) { return vf->N; }
double **valuef_get_cores(struct ValueF *vf) { return vf->cores; }

double valuef_eval_ind(struct ValueF *vf, const size_t *ind)
{
    size_t rmax = 1;
    for (size_t m = 0; m <= vf->d; m++) if (vf->ranks[m] > rmax) rmax = vf->ranks[m];
    double *v = xcalloc(2 * rmax, sizeof(double)), *w = v + rmax;
    v[0] = 1.0;
    for (size_t m = 0; m < vf->d; m++) {
        const size_t r0 = vf->ranks[m], r1 = vf->ranks[m + 1];
        const double *G = vf->cores[m] + ind[m] * r0 * r1;
        for (size_t b = 0; b < r1; b++) {
            double s = 0.0;
            for (size_t a = 0; a < r0; a++) s += v[a] * G[a + b * r0];
            w[b] = s;
        }
        memcpy(v, w, r1 * sizeof(double));
    }
    const double out = v[0];
    free(v);
    return out;
}

void c3sc_forget_ctx(struct c3sc_hip_ctx *ctx)
{ /* called before a context is destroyed: a later context at the same address must not inherit its bookkeeping */
    for (int i = 0; i < MAX_TRACKED_CTX; i++)
        if (g_ctx[i].ctx == ctx) { g_ctx[i].ctx = NULL; g_ctx[i].version = 0; g_ctx[i].cfg_set = 0; }
}

void valuef_bind_device(struct ValueF *vf, struct c3sc_hip_ctx *ctx)
{
    vf->bound = ctx;
    const int sl = ctx_slot(ctx);
    if (g_ctx[sl].version == vf->version) return;
    hipok(ctx, c3sc_hip_upload_value(ctx, vf->ranks, (const double *const *)vf->cores), "c3sc_hip_upload_value");
    g_ctx[sl].version = vf->version;
}

int valuef_eval_fiber_ind_nn(struct ValueF *vf, const size_t *fixed_ind, size_t dim_vary, const size_t *neighbors,
                             const size_t *neighbors_vary, double *out)
{
    if (vf->bound == NULL) DIE("valuef_eval_fiber_ind_nn: value function is not bound to a device (valuef_bind_device)");
    valuef_bind_device(vf, vf->bound);
    const size_t d = vf->d, N = vf->N[dim_vary];
    int32_t idx[C3SC_MAX_DIM], nbf[2 * C3SC_MAX_DIM];
    int32_t *nbv = xcalloc(2 * N, sizeof(int32_t));
    for (size_t m = 0; m < d; m++) idx[m] = (m == dim_vary) ? 0 : (int32_t)fixed_ind[m];
    for (size_t i = 0; i < 2 * (d - 1); i++) nbf[i] = (int32_t)neighbors[i];
    for (size_t i = 0; i < 2 * N; i++) nbv[i] = (int32_t)neighbors_vary[i];
    hipok(vf->bound, c3sc_hip_stencil_fibers_nb_host(vf->bound, (int)dim_vary, 1, idx, nbf, nbv, out, NULL),
          "c3sc_hip_stencil_fibers_nb_host");
    free(nbv);
    return 0;
}

/* =============================================================================== nodeutil */
static int assemble_rates(size_t dx, size_t du, size_t dw, double h2, const double *tv, const double *hvec, const double *drift,
                        const double *grad_drift, const double *ddiff, const double *grad_ddiff, double *prob,
                        double *grad_prob, double *dt, double *grad_dt, double *space, int twice_left)
{ /* nodeutil.c:267-406: upwind rates p_m^{-+} = t2_m s_mm^2/2 + t_m max(-+b_m, 0), dead zone 1e-14 */
    const int want_grad = (grad_prob != NULL);
    int res = 0;
    if (space != NULL) for (size_t j = 0; j < du; j++) space[j] = 0.0;
    double Q = 0.0;
    for (size_t m = 0; m < dx; m++) {
        /* new routine: (h^2/h_m, h^2/h_m^2) handed in (nodeutil.c:291-292); old one: built from the spacings (:126-127) */
        const double t = tv ? tv[2 * m] : h2 / hvec[m];
        const double t2 = tv ? tv[2 * m + 1] : t / hvec[m];
        const double s2 = ddiff[m * dx + m] * ddiff[m * dx + m];
        const double base = t2 * s2 / 2.0;
        const int sgn = drift[m] < -1e-14 ? -1 : (drift[m] > 1e-14 ? 1 : 0);
        double *pm = &prob[2 * m], *pp = &prob[2 * m + 1];
        *pm = base;
        *pp = base;
        if (sgn < 0) *pm -= t * drift[m];
        if (sgn > 0) *pp += t * drift[m];
        Q += *pm;
        Q += *pp;
        if (want_grad) {
            double *gm = grad_prob + 2 * m * du, *gp = gm + du;
            for (size_t j = 0; j < du; j++) gm[j] = gp[j] = t2 * grad_ddiff[m * dx + m + j * dx * dw];
            if (sgn < 0) { for (size_t j = 0; j < du; j++) gm[j] += -t * grad_drift[m + j * dx]; }
            else if (sgn > 0) { for (size_t j = 0; j < du; j++) gp[j] += t * grad_drift[m + j * dx]; }
            else {
                for (size_t j = 0; j < du; j++) {
                    const double g = grad_drift[j * dx + m];
                    if (g < 0) { /* the old routine applies this update twice, a daxpy and a loop (nodeutil.c:162-165) */
                        for (int rep = 0; rep <= twice_left; rep++)
                            for (size_t q = 0; q < du; q++) gm[q] -= t * grad_drift[m + q * dx];
                    }
                    else if (g > 0) { for (size_t q = 0; q < du; q++) gp[q] += t * grad_drift[m + q * dx]; }
                    else res = 2;
                }
            }
            for (size_t j = 0; j < du; j++) space[j] += gm[j];
            for (size_t j = 0; j < du; j++) space[j] += gp[j];
        }
    }
    if (Q < 1e-14) return 1; /* stationary: dt, normalised probabilities are not produced (nodeutil.c:365-367) */
    *dt = h2 / Q;
    prob[2 * dx] = 1.0;
    if (want_grad) {
        const double Q2 = Q * Q, c = h2 / Q2;
        for (size_t j = 0; j < du; j++) { grad_prob[2 * dx * du + j] = 0.0; grad_dt[j] = -space[j] * c; }
        for (size_t i = 0; i < 2 * dx; i++) {
            for (size_t j = 0; j < du; j++) grad_prob[i * du + j] = (Q * grad_prob[i * du + j] - space[j] * prob[i]) / Q2;
            prob[i] /= Q;
            prob[2 * dx] -= prob[i];
            for (size_t j = 0; j < du; j++) grad_prob[2 * dx * du + j] -= grad_prob[i * du + j];
        }
    } else {
        for (size_t i = 0; i < 2 * dx; i++) { prob[i] /= Q; prob[2 * dx] -= prob[i]; }
    }
    return res;
}

int transition_assemble(size_t dx, size_t du, size_t dw, double h2, const double *tv, const double *drift,
                        const double *grad_drift, const double *ddiff, const double *grad_ddiff, double *prob,
                        double *grad_prob, double *dt, double *grad_dt, double *space)
{ /* nodeutil.c:267-406 */
    return assemble_rates(dx, du, dw, h2, tv, NULL, drift, grad_drift, ddiff, grad_ddiff, prob, grad_prob, dt, grad_dt, space, 0);
}

int transition_assemble_old(size_t dx, size_t du, size_t dw, double h, const double *hvec, const double *drift,
                            const double *grad_drift, const double *ddiff, const double *grad_ddiff, double *prob,
                            double *grad_prob, double *dt, double *grad_dt, double *space)
{ /* nodeutil.c:82-233: h is the minimum spacing, hvec the per-dimension spacings */
    return assemble_rates(dx, du, dw, h * h, NULL, hvec, drift, grad_drift, ddiff, grad_ddiff, prob, grad_prob, dt, grad_dt, space, 1);
}

static size_t find_node(double x, size_t N, const double *grid)
{ /* O(N) scan, absolute tolerance 1e-14 (nodeutil.c:408-419, quirk Q9) */
    size_t i = 0;
    while (i < N && !(fabs(x - grid[i]) < 1e-14)) i++;
    return i;
}

int convert_fiber_to_ind(size_t d, size_t N, const double *x, const size_t *Ngrid, double **xgrid, size_t *fixed_ind,
                         size_t *dim_vary)
{ /* nodeutil.c:437-470: node 0 gives the indices, node 1 tells which coordinate moves */
    for (size_t m = 0; m < d; m++) {
        fixed_ind[m] = find_node(x[m], Ngrid[m], xgrid[m]);
        if (fixed_ind[m] == Ngrid[m]) {
            fprintf(stderr, "Error: evaluation is not on the grid\nx[%zu] = %3.15E\n", m, x[m]);
            return 1;
        }
    }
    *dim_vary = d;
    for (size_t m = 0; m < d && *dim_vary == d; m++)
        if (find_node(x[d + m], Ngrid[m], xgrid[m]) != fixed_ind[m]) *dim_vary = m;
    if (*dim_vary == d) return 1;
    return (N != Ngrid[*dim_vary]) ? 2 : 0;
}

int process_fibers_neighbor(size_t d, const size_t *fixed_ind, size_t dim_vary, const double *x, int *absorbed,
                            size_t *nv, size_t *nf, const size_t *ngrid, const struct Boundary *bound)
{ /* nodeutil.c:489-627 */
    const size_t N = ngrid[dim_vary];
    int whole = 0;
    size_t *o = nf;
    for (size_t m = 0; m < d; m++) {
        if (m == dim_vary) continue;
        const size_t i = fixed_ind[m], last = ngrid[m] - 1;
        const enum EBTYPE b = boundary_type_dim(bound, m, i == 0 ? 0 : 1);
        o[0] = i - 1;
        o[1] = i + 1;
        if (i == 0 || i == last) {
            if (b == ABSORB) { o[0] = o[1] = i; whole = 1; }
            else if (b == REFLECT) { if (i == 0) o[0] = i; else o[1] = i; }
            else if (b == PERIODIC) { if (i == 0) o[0] = ngrid[m] - 2; else o[1] = 1; }
            else { fprintf(stderr, "No boundary specified!\n"); assert(0); }
        }
        o += 2;
    }
    for (size_t j = 0; j < N; j++) absorbed[j] = whole ? 1 : (boundary_in_obstacle(bound, x + j * d) ? -1 : 0);
    /* end points follow dim_vary's own boundary type and overwrite the flags above (quirk Q3) */
    const enum EBTYPE b = boundary_type_dim(bound, dim_vary, 0);
    if (b != ABSORB && b != REFLECT && b != PERIODIC) { fprintf(stderr, "Should not be here!\n"); assert(0); }
    if (b == ABSORB) absorbed[0] = absorbed[N - 1] = 1;
    else if (!boundary_get_consistent_ends(bound)) absorbed[0] = absorbed[N - 1] = 0; /* else: boundary_set_consistent_ends keeps them */
    for (size_t j = 0; j < N; j++) {
        size_t lo = j, hi = j;
        if (j == 0) { if (b == REFLECT) hi = 1; else if (b == PERIODIC) { lo = N - 2; hi = 1; } }
        else if (j == N - 1) { if (b == REFLECT) lo = N - 2; else if (b == PERIODIC) { lo = N - 2; hi = 1; } }
        else if (absorbed[j] == 0) { lo = j - 1; hi = j + 1; }
        nv[2 * j] = lo;
        nv[2 * j + 1] = hi;
    }
    return 0;
}

int mca_get_neighbor_costs(size_t d, size_t N, const double *x, struct Boundary *bound, struct ValueF *vf,
                           const size_t *ngrid, double **xgrid, size_t *fixed_ind, size_t *dim_vary, int *absorbed,
                           double *out)
{ /* nodeutil.c:647-713 */
    int res = convert_fiber_to_ind(d, N, x, ngrid, xgrid, fixed_ind, dim_vary);
    if (res != 0) { printf("\n======================================\nError calling convert fiber to _ind!!\n"); }
    assert(res == 0);
    size_t *nv = xcalloc(2 * N, sizeof(size_t)), *nf = xcalloc(2 * d, sizeof(size_t));
    res = process_fibers_neighbor(d, fixed_ind, *dim_vary, x, absorbed, nv, nf, ngrid, bound);
    assert(res == 0);
    res = valuef_eval_fiber_ind_nn(vf, fixed_ind, *dim_vary, nf, nv, out);
    free(nv);
    free(nf);
    return res;
}

/* =============================================================================== bellman */
double bellmanrhs(size_t dx, size_t du, double stage_cost, const double *stage_grad, double discount,
                  const double *prob, const double *prob_grad, double dt, const double *dtgrad, const double *cost,
                  double *grad)
{ /* bellman.c:88-112 */
    const double ebt = exp(-discount * dt);
    double ctg = 0.0;
    for (size_t i = 0; i < 2 * dx + 1; i++) ctg += prob[i] * cost[i];
    if (grad != NULL) {
        for (size_t j = 0; j < du; j++) {
            double g = stage_grad[j] * dt + dtgrad[j] * stage_cost;
            g += (-discount) * dtgrad[j] * ebt * ctg;
            for (size_t i = 0; i < 2 * dx + 1; i++) g += ebt * prob_grad[i * du + j] * cost[i];
            grad[j] = g;
        }
    }
    return dt * stage_cost + ebt * ctg;
}

struct MCAparam { size_t dx, du; size_t *ngrid; double **xgrid; double hmin, *hvec, h2, *t; };

struct MCAparam *mca_param_create(size_t dx, size_t du)
{
    struct MCAparam *m = xcalloc(1, sizeof(*m));
    m->dx = dx; m->du = du;
    return m;
}

void mca_add_grid_refs(struct MCAparam *m, size_t *ngrid, double **xgrid, double hmin, double *hvec)
{ /* borrows the grid; t[2i] = h2/h_i, t[2i+1] = h2/h_i^2 (bellman.c:181-186) */
    m->ngrid = ngrid; m->xgrid = xgrid; m->hmin = hmin; m->hvec = hvec;
    m->h2 = hmin * hmin;
    free(m->t);
    m->t = xcalloc(2 * m->dx, sizeof(double));
    for (size_t i = 0; i < m->dx; i++) { m->t[2 * i] = m->h2 / hvec[i]; m->t[2 * i + 1] = m->t[2 * i] / hvec[i]; }
}

void mca_param_destroy(struct MCAparam *m) { if (m) { free(m->t); free(m); } }

struct DPparam {
    struct Drift *drift;
    struct Diff *diff;
    struct Boundary *bound;
    double discount;
    int (*stagecost)(double, const double *, const double *, double *, double *);
    int (*boundcost)(double, const double *, double *);
    int (*obscost)(const double *, double *);
    int model;
    double prm[C3SC_MAX_PARAMS];
    int nprm;
    int model_checked;
};

struct DPparam *dp_param_create(size_t dx, size_t du, size_t dw, double discount)
{
    struct DPparam *dp = xcalloc(1, sizeof(*dp));
    dp->drift = drift_alloc(dx, du);
    dp->diff = diff_alloc(dx, du, dw);
    dp->discount = discount;
    return dp;
}
void dp_param_destroy(struct DPparam *dp) { if (dp) { drift_free(dp->drift); diff_free(dp->diff); free(dp); } }
void dp_param_add_drift(struct DPparam *dp, c3sc_dyn_fn b, void *a) { drift_add_func(dp->drift, b, a); }
void dp_param_add_diff(struct DPparam *dp, c3sc_dyn_fn s, void *a) { diff_add_func(dp->diff, s, a); }
void dp_param_add_boundary(struct DPparam *dp, struct Boundary *b) { dp->bound = b; }
void dp_param_add_stagecost(struct DPparam *dp, int (*f)(double, const double *, const double *, double *, double *)) { dp->stagecost = f; }
void dp_param_add_boundcost(struct DPparam *dp, int (*f)(double, const double *, double *)) { dp->boundcost = f; }
void dp_param_add_obscost(struct DPparam *dp, int (*f)(const double *, double *)) { dp->obscost = f; }
void dp_param_set_device_model(struct DPparam *dp, int model, const double *params, size_t nparams)
{
    if (nparams > C3SC_MAX_PARAMS) DIE("dp_param_set_device_model: too many parameters");
    dp->model = model;
    dp->nprm = (int)nparams;
    memset(dp->prm, 0, sizeof(dp->prm));
    for (size_t i = 0; i < nparams; i++) dp->prm[i] = params[i];
    dp->model_checked = 0;
}

struct ControlParams {
    double time;
    size_t dx, dw, N;
    const double *x; /* borrowed (bellman.c:329-335) */
    struct DPparam *dp;
    struct MCAparam *mca;
    struct Workspace *work;
    struct c3Opt *opt;
    int res_last_grad;
};

struct ControlParams *control_params_create(size_t dx, size_t dw, struct DPparam *dp, struct MCAparam *mca,
                                            struct Workspace *work, struct c3Opt *opt)
{
    struct ControlParams *c = xcalloc(1, sizeof(*c));
    c->dx = dx; c->dw = dw; c->dp = dp; c->mca = mca; c->work = work; c->opt = opt;
    return c;
}
void control_params_add_time_and_states(struct ControlParams *c, double time, size_t N, const double *x) { c->time = time; c->N = N; c->x = x; }
int control_params_get_last_res(const struct ControlParams *c) { return c->res_last_grad; }
void control_params_destroy(struct ControlParams *c) { free(c); }

struct Memory { void *shared; size_t private; }; /* bellman.c:59-63 */

double bellman_control(size_t du, const double *u, double *grad_u, void *args)
{ /* bellman.c:367-480: objective of one control at one node; reads the node's costs/absorbed from the workspace */
    struct Memory *mem = args;
    struct ControlParams *p = mem->shared;
    const size_t node = mem->private, dx = p->dx, dw = p->dw;
    struct DPparam *dp = p->dp;
    struct Workspace *w = p->work;
    const double *x = p->x + node * dx;
    const int ab = *workspace_get_absorbed(w, node);
    double val = 0.0;
    if (grad_u != NULL) for (size_t i = 0; i < du; i++) grad_u[i] = 0.0;
    if (ab == 1) { int r = dp->boundcost(p->time, x, &val); assert(r == 0); (void)r; return val; }
    if (ab == -1) { int r = dp->obscost(x, &val); assert(r == 0); (void)r; return val; }
    if (ab != 0) { fprintf(stderr, "Unrecognized aborbed condition %d\n", ab); exit(1); }
    double *drift = workspace_get_drift(w, node), *diff = workspace_get_diff(w, node);
    double *prob = workspace_get_prob(w, node), *dt = workspace_get_dt(w, node);
    double *costs = workspace_get_costs(w, node);
    double stage;
    int res;
    if (grad_u != NULL) {
        double *gdrift = workspace_get_grad_drift(w, node), *gdiff = workspace_get_grad_diff(w, node);
        double *gprob = workspace_get_grad_prob(w, node), *gdt = workspace_get_grad_dt(w, node);
        double *gstage = workspace_get_grad_stage(w, node), *space = workspace_get_control_size_extra(w, node);
        res = drift_eval(dp->drift, p->time, x, u, drift, gdrift); assert(res == 0);
        res = diff_eval(dp->diff, p->time, x, u, diff, gdiff); assert(res == 0);
        res = dp->stagecost(p->time, x, u, &stage, gstage); assert(res == 0);
        res = transition_assemble(dx, du, dw, p->mca->h2, p->mca->t, drift, gdrift, diff, gdiff, prob, gprob, dt, gdt, space);
        p->res_last_grad = res;
        val = bellmanrhs(dx, du, stage, gstage, dp->discount, prob, gprob, *dt, gdt, costs, grad_u);
    } else {
        res = drift_eval(dp->drift, p->time, x, u, drift, NULL); assert(res == 0);
        res = diff_eval(dp->diff, p->time, x, u, diff, NULL); assert(res == 0);
        res = dp->stagecost(p->time, x, u, &stage, NULL); assert(res == 0);
        res = transition_assemble(dx, du, dw, p->mca->h2, p->mca->t, drift, NULL, diff, NULL, prob, NULL, dt, NULL, NULL);
        assert(res == 0); /* bellman.c:452 */
        val = bellmanrhs(dx, du, stage, NULL, dp->discount, prob, NULL, *dt, NULL, costs, NULL);
    }
    (void)res;
    return val;
}

int bellman_optimal(size_t du, double *u, double *val, void *arg)
{ /* bellman.c:504-543 (BRUTEFORCE) and the box branch of :545-1118: single node on the host (c3control_controller) */
    struct Memory *mem = arg;
    struct ControlParams *p = mem->shared;
    const int ab = *workspace_get_absorbed(p->work, mem->private);
    if (ab != 0) {
        for (size_t i = 0; i < du; i++) u[i] = 0.0;
        *val = bellman_control(du, u, NULL, arg);
        return 0;
    }
    struct c3Opt *opt = c3opt_copy(p->opt);
    c3opt_add_objective(opt, &bellman_control, mem);
    c3opt_minimize(opt, u, val); /* BRUTEFORCE: list scan; otherwise the box minimiser (grid + golden section) */
    c3opt_free(opt);
    return 0;
}

struct VIparam {
    struct ControlParams *cp;
    struct ValueF *vf;
    size_t nstate_evals, nnode_evals;
    double convergence, time_in_loop;
};

struct VIparam *vi_param_create(double convergence)
{
    struct VIparam *vi = xcalloc(1, sizeof(*vi));
    vi->convergence = convergence;
    return vi;
}
void vi_param_destroy(struct VIparam *vi) { free(vi); }
void vi_param_add_cp(struct VIparam *vi, struct ControlParams *cp) { vi->cp = cp; }
void vi_param_add_value(struct VIparam *vi, struct ValueF *vf) { vi->vf = vf; vi->nstate_evals = 0; vi->nnode_evals = 0; }
size_t vi_param_get_nnode_evals(const struct VIparam *vi) { return vi->nnode_evals; }

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}

/* push the host-side problem description to a device context (cheap; idempotent) and make vf resident on it */
static struct c3sc_hip_ctx *sync_device_ctx(struct ControlParams *cp, struct c3sc_hip_ctx *ctx, struct ValueF *vf)
{
    struct MCAparam *mca = cp->mca;
    struct DPparam *dp = cp->dp;
    const size_t d = mca->dx;
    if (dp->model == 0 && (dp->stagecost == NULL || dp->boundcost == NULL || dp->obscost == NULL))
        DIE("bellman_vi/pi: neither a device model (dp_param_set_device_model) nor the host callbacks are set");
    const int brute = c3opt_is_bruteforce(cp->opt);
    const size_t odu = c3opt_get_d(cp->opt);
    if (!brute && odu > C3SC_MAX_DU) DIE("bellman_vi/pi: the box minimiser handles up to %d control dimensions", C3SC_MAX_DU);
    if (!brute && dp->model == 0) DIE("bellman_vi/pi: continuous controls need a device model (dp_param_set_device_model); host-callback tables are per candidate");
    /* signature of everything the device holds besides the value function */
    int bc[C3SC_MAX_DIM];
    double lb[C3SC_MAX_OBSTACLES * C3SC_MAX_DIM], ub[C3SC_MAX_OBSTACLES * C3SC_MAX_DIM];
    const size_t nobs = boundary_get_nobs(dp->bound);
    uint64_t sig = 1469598103934665603ULL;
    for (size_t m = 0; m < d; m++) {
        bc[m] = (int)boundary_type_dim(dp->bound, m, 0);
        sig = fnv(sig, &bc[m], sizeof(int));
        sig = fnv(sig, &mca->ngrid[m], sizeof(size_t));
        sig = fnv(sig, mca->xgrid[m], mca->ngrid[m] * sizeof(double));
    }
    for (size_t o = 0; o < nobs; o++)
        for (size_t m = 0; m < d; m++) {
            lb[o * d + m] = boundary_obstacle_get_lb(dp->bound, o)[m];
            ub[o * d + m] = boundary_obstacle_get_ub(dp->bound, o)[m];
        }
    sig = fnv(sig, lb, nobs * d * sizeof(double));
    sig = fnv(sig, ub, nobs * d * sizeof(double));
    sig = fnv(sig, mca->t, 2 * d * sizeof(double));
    sig = fnv(sig, &mca->h2, sizeof(double));
    sig = fnv(sig, &dp->discount, sizeof(double));
    sig = fnv(sig, &dp->model, sizeof(int));
    sig = fnv(sig, dp->prm, sizeof(dp->prm));
    sig = fnv(sig, &brute, sizeof(int));
    const int cends = boundary_get_consistent_ends(dp->bound);
    sig = fnv(sig, &cends, sizeof(int));
    if (brute) sig = fnv(sig, c3opt_get_brute_vals(cp->opt), c3opt_get_nbrute(cp->opt) * odu * sizeof(double));
    else {
        const size_t g = c3opt_get_box_grid(cp->opt), pl = c3opt_get_box_polish(cp->opt);
        sig = fnv(sig, c3opt_get_lb(cp->opt), odu * sizeof(double));
        sig = fnv(sig, c3opt_get_ub(cp->opt), odu * sizeof(double));
        sig = fnv(sig, &g, sizeof(g));
        sig = fnv(sig, &pl, sizeof(pl));
    }
    const int sl = ctx_slot(ctx);
    if (!g_ctx[sl].cfg_set || g_ctx[sl].cfg_sig != sig) {
        hipok(ctx, c3sc_hip_set_grid(ctx, (int)d, mca->ngrid, (const double *const *)mca->xgrid), "c3sc_hip_set_grid");
        hipok(ctx, c3sc_hip_set_boundary(ctx, bc, (int)nobs, lb, ub), "c3sc_hip_set_boundary");
        hipok(ctx, c3sc_hip_set_mca(ctx, mca->h2, mca->t, dp->discount), "c3sc_hip_set_mca");
        hipok(ctx, c3sc_hip_set_consistent_ends(ctx, cends), "c3sc_hip_set_consistent_ends");
        if (dp->model != 0) hipok(ctx, c3sc_hip_set_model(ctx, dp->model, dp->prm, dp->nprm), "c3sc_hip_set_model");
        if (brute)
            hipok(ctx, c3sc_hip_set_controls(ctx, (int)c3opt_get_nbrute(cp->opt), (int)odu, c3opt_get_brute_vals(cp->opt)), "c3sc_hip_set_controls");
        else
            hipok(ctx, c3sc_hip_set_control_box(ctx, (int)odu, c3opt_get_lb(cp->opt), c3opt_get_ub(cp->opt), (int)c3opt_get_box_grid(cp->opt),
                                                (int)c3opt_get_box_polish(cp->opt)), "c3sc_hip_set_control_box");
        g_ctx[sl].cfg_set = 1;
        g_ctx[sl].cfg_sig = sig;
        g_ctx[sl].version = 0; /* set_grid invalidates the resident value function */
    }
    valuef_bind_device(vf, ctx);
    return ctx;
}

static struct c3sc_hip_ctx *sync_device(struct VIparam *vi)
{
    return sync_device_ctx(vi->cp, workspace_get_hip_ctx(vi->cp->work), vi->vf);
}

/* first-use cross-check of the device model against the host callbacks: a handful of live nodes are
 * re-evaluated on the host with bellman_optimal (callbacks + transition_assemble + bellmanrhs). */
static void cross_check_model(struct VIparam *vi, struct c3sc_hip_ctx *ctx, size_t k, const int32_t *idx, size_t N,
                              const double *x, const double *gpu_out, const int32_t *gpu_abs)
{
    struct ControlParams *cp = vi->cp;
    struct DPparam *dp = cp->dp;
    if (dp->model_checked || dp->stagecost == NULL || dp->boundcost == NULL || dp->obscost == NULL) return;
    const size_t dx = cp->dx, S = 2 * dx + 1;
    double *costs = xcalloc(N * S, sizeof(double));
    int32_t *ab = xcalloc(N, sizeof(int32_t));
    hipok(ctx, c3sc_hip_stencil_fibers_host(ctx, (int)k, 1, idx, costs, ab), "c3sc_hip_stencil_fibers_host");
    control_params_add_time_and_states(cp, 0.0, N, x);
    double u[16];
    size_t checked = 0;
    for (size_t j = 0; j < N && checked < 6; j += (N > 6 ? N / 6 : 1)) {
        if (ab[j] != gpu_abs[j]) DIE("device/host absorbed flag mismatch at node %zu", j);
        memcpy(workspace_get_costs(cp->work, 0), costs + j * S, S * sizeof(double));
        *workspace_get_absorbed(cp->work, 0) = ab[j];
        struct Memory mem = {cp, 0};
        const double *xs = cp->x;
        cp->x = x + j * dx; /* node 0 of a one-node view */
        double hv;
        bellman_optimal(c3opt_get_d(cp->opt), u, &hv, &mem);
        cp->x = xs;
        const double tol = 1e-9 * (fabs(hv) > 1.0 ? fabs(hv) : 1.0);
        if (fabs(hv - gpu_out[j]) > tol)
            DIE("device model %d does not reproduce the host callbacks: node %zu host %.17g device %.17g", dp->model, j, hv, gpu_out[j]);
        checked++;
    }
    dp->model_checked = 1;
    free(costs);
    free(ab);
}

static int dp_has_device_model(const struct DPparam *dp) { return dp->model != 0; }

/* host evaluation of the user's callbacks for one fiber -> [N][U][2dx+1] and [N][2] */
static void eval_callback_tables(struct ControlParams *cp, size_t k, const int32_t *idx, size_t N, const double *x,
                                 double *tables, double *costs2)
{
    struct DPparam *dp = cp->dp;
    const size_t dx = cp->dx, dw = cp->dw, du = c3opt_get_d(cp->opt), U = c3opt_get_nbrute(cp->opt), S = 2 * dx + 1;
    const double *cands = c3opt_get_brute_vals(cp->opt);
    size_t fi[C3SC_MAX_DIM], nf[2 * C3SC_MAX_DIM];
    size_t *nv = xcalloc(2 * N, sizeof(size_t));
    int *ab = xcalloc(N, sizeof(int));
    double *drift = xcalloc(dx, sizeof(double)), *diff = xcalloc(dx * dw, sizeof(double));
    for (size_t m = 0; m < dx; m++) fi[m] = (size_t)idx[m];
    process_fibers_neighbor(dx, fi, k, x, ab, nv, nf, cp->mca->ngrid, dp->bound);
    for (size_t j = 0; j < N; j++) {
        const double *xj = x + j * dx;
        int res = 0;
        if (ab[j] == 1) res = dp->boundcost(cp->time, xj, &costs2[2 * j]);
        else if (ab[j] == -1) res = dp->obscost(xj, &costs2[2 * j + 1]);
        else {
            for (size_t c = 0; c < U && res == 0; c++) {
                double *row = tables + (j * U + c) * S;
                res = drift_eval(dp->drift, cp->time, xj, cands + c * du, drift, NULL);
                if (res == 0) res = diff_eval(dp->diff, cp->time, xj, cands + c * du, diff, NULL);
                if (res == 0) res = dp->stagecost(cp->time, xj, cands + c * du, &row[2 * dx], NULL);
                for (size_t m = 0; m < dx; m++) { row[m] = drift[m]; row[dx + m] = diff[m * dx + m]; } /* nodeutil.c:294 */
            }
        }
        assert(res == 0);
        (void)res;
    }
    free(nv); free(ab); free(drift); free(diff);
}

/* coordinates of the nodes of F fibers from their grid indices (what the cross approximation would hand over) */
static double *fibers_x_from_idx(const struct MCAparam *mca, size_t F, size_t k, const int32_t *idx)
{
    const size_t dx = mca->dx, N = mca->ngrid[k];
    double *x = xcalloc(F * N * dx, sizeof(double));
    for (size_t f = 0; f < F; f++)
        for (size_t j = 0; j < N; j++)
            for (size_t m = 0; m < dx; m++)
                x[(f * N + j) * dx + m] = (m == k) ? mca->xgrid[m][j] : mca->xgrid[m][idx[f * dx + m]];
    return x;
}

/* bellman.c:1295-1423 for F fibers along dim k0 given by their grid indices: memo per node, every fiber with a
 * missing node in ONE launch.  fast = 0: the reference's string-keyed table; fast = 1: its integer-keyed twin. */
/* C3SC_PROFILE=1: where a sweep's host time goes (printed by c3control_end_vi) */
#include <time.h>
static double g_t_lookup, g_t_device, g_t_store;
static size_t g_n_calls;
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* transition_assemble's "stationary node" outcome (nodeutil.c:365-367; bellman.c:452 asserts on it) comes back as a
 * device flag.  The public per-fiber entry points read it after every call; the index-based batch entry points of the
 * solver loops read it once per sweep (c3control_end_vi / end_pi_step) -- a status read is a device round trip. */
static void die_if_stationary(struct c3sc_hip_ctx *ctx)
{
    unsigned st = 0;
    hipok(ctx, c3sc_hip_get_status(ctx, &st, 1), "c3sc_hip_get_status");
    if (st & C3SC_STATUS_STATIONARY) DIE("transition_assemble: stationary node (Q < 1e-14); the reference asserts here (bellman.c:452)");
}

int c3sc_memo_bypass = 0; /* diagnostic: the index-based entry points neither look up nor store (every node is evaluated) */

static int vi_core(struct VIparam *vi, size_t F, size_t k0, const int32_t *idx, const double *x_in, double *out, int fast)
{
    struct ControlParams *cp = vi->cp;
    struct MCAparam *mca = cp->mca;
    const size_t dx = mca->dx, N = mca->ngrid[k0];
    struct c3sc_hip_ctx *ctx = sync_device(vi); /* first call: creates the device context and uploads the cores */
    const double t_begin = now_s();
    struct HTable *ht = workspace_get_vi_htable(cp->work);
    struct FastMemo *fm = workspace_get_vi_fastmemo(cp->work);
    size_t *ser = workspace_get_ind_to_serialize(cp->work);
    const size_t vi_iter = workspace_get_vi_iter(cp->work);
    char key[256];
    struct FmFiber ff;
    unsigned char *need = xcalloc(F, 1), *hit = xcalloc(F * N, 1);
    size_t *slot = fast ? xcalloc(F * N, sizeof(size_t)) : NULL; /* where a missing node will be stored (fastmemo_fiber_put_at) */
    const size_t cap_then = fast ? fastmemo_cap(fm) : 0;
    for (size_t f = 0; f < F; f++) {
        for (size_t m = 0; m < dx; m++) ser[m] = (size_t)idx[f * dx + m];
        ser[dx] = 0;           /* bellman.c:1337 */
        ser[dx + 1] = vi_iter; /* bellman.c:1338 */
        if (fast) { fastmemo_fiber_begin(&ff, dx, idx + f * dx, k0, 0, vi_iter); fastmemo_fiber_prefetch(fm, &ff, N); }
        for (size_t j = 0; j < N; j++) { /* memo lookup, bellman.c:1341-1353 */
            double v = 0.0;
            int found;
            if (fast) {
                found = fastmemo_fiber_get_slot(fm, &ff, j, &v, &slot[f * N + j]);
                if (c3sc_memo_bypass) found = 0;
            } else {
                ser[k0] = j;
                size_t_a_to_char(ser, dx + 2, key);
                size_t nb = 0;
                double *pv = htable_get_element(ht, key, &nb);
                found = pv != NULL;
                if (found) v = pv[0];
            }
            if (found) { out[f * N + j] = v; hit[f * N + j] = 1; }
            else need[f] = 1;
        }
    }
    /* compact the fibers that still need work and run them in one launch */
    size_t nrun = 0;
    for (size_t f = 0; f < F; f++) nrun += need[f];
    const double t_looked = now_s();
    g_t_lookup += t_looked - t_begin;
    g_n_calls++;
    if (nrun > 0) {
        int32_t *ridx = xcalloc(nrun * dx, sizeof(int32_t));
        double *rout = xcalloc(nrun * N, sizeof(double));
        /* absorbed flags travel back only while the first-fiber cross-check is still pending */
        const int want_abs = dp_has_device_model(cp->dp) && !cp->dp->model_checked && cp->dp->stagecost != NULL &&
                             cp->dp->boundcost != NULL && cp->dp->obscost != NULL;
        int32_t *rabs = want_abs ? xcalloc(nrun * N, sizeof(int32_t)) : NULL;
        size_t r = 0;
        for (size_t f = 0; f < F; f++)
            if (need[f]) memcpy(ridx + (r++) * dx, idx + f * dx, dx * sizeof(int32_t));
        double *x_own = NULL;
        const int want_x = !dp_has_device_model(cp->dp) ||
                           (!cp->dp->model_checked && cp->dp->stagecost != NULL && cp->dp->boundcost != NULL && cp->dp->obscost != NULL);
        const double *x = x_in;
        if (x == NULL && want_x) { x_own = fibers_x_from_idx(mca, F, k0, idx); x = x_own; }
        if (!c3opt_is_bruteforce(cp->opt)) {
            hipok(ctx, c3sc_hip_bellman_fibers_box_host(ctx, (int)k0, nrun, ridx, rout, NULL, rabs), "c3sc_hip_bellman_fibers_box_host");
        } else if (dp_has_device_model(cp->dp)) {
            hipok(ctx, c3sc_hip_bellman_fibers_host(ctx, (int)k0, nrun, ridx, rout, NULL, rabs), "c3sc_hip_bellman_fibers_host");
        } else {
            /* universal path: the user's callbacks are evaluated here, exactly where bellman_control would call
             * them (bellman.c:414-444, 458, 467), and shipped as tables */
            const size_t U = c3opt_get_nbrute(cp->opt), S = 2 * dx + 1;
            double *tables = xcalloc(nrun * N * U * S, sizeof(double));
            double *costs2 = xcalloc(nrun * N * 2, sizeof(double));
            size_t rr = 0;
            for (size_t f = 0; f < F; f++)
                if (need[f]) { eval_callback_tables(cp, k0, idx + f * dx, N, x + f * N * dx, tables + rr * N * U * S, costs2 + rr * N * 2); rr++; }
            hipok(ctx, c3sc_hip_bellman_fibers_tables_host(ctx, (int)k0, nrun, ridx, tables, costs2, rout, NULL, rabs),
                  "c3sc_hip_bellman_fibers_tables_host");
            free(tables);
            free(costs2);
        }
        if (!fast) die_if_stationary(ctx);
        const double t_dev = now_s();
        g_t_device += t_dev - t_looked;
        r = 0;
        for (size_t f = 0; f < F; f++) {
            if (!need[f]) continue;
            if (r == 0 && want_abs && x != NULL) cross_check_model(vi, ctx, k0, ridx, N, x + f * N * dx, rout, rabs);
            for (size_t m = 0; m < dx; m++) ser[m] = (size_t)idx[f * dx + m];
            ser[dx] = 0;
            ser[dx + 1] = vi_iter;
            if (fast) { fastmemo_fiber_begin(&ff, dx, idx + f * dx, k0, 0, vi_iter); fastmemo_fiber_prefetch(fm, &ff, N); }
            for (size_t j = 0; j < N; j++) {
                if (hit[f * N + j]) continue;
                out[f * N + j] = rout[r * N + j];
                if (fast) {
                    if (!c3sc_memo_bypass) fastmemo_fiber_put_at(fm, &ff, j, out[f * N + j], slot[f * N + j], cap_then);
                } else {
                    ser[k0] = j;
                    size_t_a_to_char(ser, dx + 2, key);
                    htable_add_element(ht, key, &out[f * N + j], 1); /* bellman.c:1383, 1413-1417 */
                }
                vi->nstate_evals++;
                vi->nnode_evals++;
            }
            r++;
        }
        free(ridx); free(rout); free(rabs); free(x_own);
        g_t_store += now_s() - t_dev;
    }
    free(need); free(hit); free(slot);
    return 0;
}

int bellman_vi_batch(size_t F, size_t N, const double *x, double *out, void *arg)
{ /* the reference's coordinate interface: recover the grid indices (nodeutil.c:437-470), then the common core */
    struct VIparam *vi = arg;
    struct ControlParams *cp = vi->cp;
    assert(cp != NULL && vi->vf != NULL);
    struct MCAparam *mca = cp->mca;
    const size_t dx = mca->dx;
    int32_t *idx = xcalloc(F * dx, sizeof(int32_t));
    size_t fi[C3SC_MAX_DIM], k0 = dx;
    for (size_t f = 0; f < F; f++) {
        size_t dv;
        int res = convert_fiber_to_ind(dx, N, x + f * N * dx, mca->ngrid, mca->xgrid, fi, &dv);
        if (res != 0) { printf("\n======================================\nError calling convert fiber to _ind!!\n"); }
        assert(res == 0 && dv < dx && N == mca->ngrid[dv]); /* nodeutil.c:681-683 */
        if (k0 == dx) k0 = dv;
        if (dv != k0) DIE("bellman_vi_batch: all fibers of a batch must vary the same dimension");
        for (size_t m = 0; m < dx; m++) idx[f * dx + m] = (m == dv) ? 0 : (int32_t)fi[m];
    }
    const int rc = (F > 0) ? vi_core(vi, F, k0, idx, x, out, 0) : 0;
    free(idx);
    return rc;
}

/* new: the same for fibers given by grid indices (idx[F][dx], entry k ignored) -- what the own cross driver
 * calls: no coordinate round trip, integer-keyed memo */
int bellman_vi_batch_idx(size_t F, size_t k, const int32_t *idx, double *out, void *arg)
{
    struct VIparam *vi = arg;
    assert(vi->cp != NULL && vi->vf != NULL && k < vi->cp->mca->dx);
    return (F > 0) ? vi_core(vi, F, k, idx, NULL, out, 1) : 0;
}

int bellman_vi(size_t N, const double *x, double *out, void *arg) { return bellman_vi_batch(1, N, x, out, arg); }

/* sharded core steps (c3control_set_fiber_sharding): the rows other ranks evaluated enter this rank's memo as if they had
 * been evaluated here -- first value stays, as in the reference's table (bellman.c:1349-1353, hashgrid.c:252-261) */
static void vi_absorb_foreign(size_t F, size_t k, const int32_t *idx, const double *out, size_t lo, size_t hi, void *arg)
{
    struct VIparam *vi = arg;
    struct ControlParams *cp = vi->cp;
    const size_t dx = cp->mca->dx, N = cp->mca->ngrid[k];
    struct FastMemo *fm = workspace_get_vi_fastmemo(cp->work);
    const size_t vi_iter = workspace_get_vi_iter(cp->work);
    struct FmFiber ff;
    if (c3sc_memo_bypass) return;
    for (size_t f = 0; f < F; f++) {
        if (f >= lo && f < hi) continue;
        fastmemo_fiber_begin(&ff, dx, idx + f * dx, k, 0, vi_iter);
        for (size_t j = 0; j < N; j++) fastmemo_fiber_put(fm, &ff, j, out[f * N + j]);
    }
}


/* =============================================================================== policy iteration */
struct PIparam { /* bellman.c:1430-1444 */
    struct ControlParams *cp;
    struct ValueF *vf_iteration;
    struct ValueF *vf_policy;
    size_t npol_evals;       /* nodes whose policy was computed */
    size_t niter_evals;      /* fiber nodes evaluated in this step */
    size_t niter_node_evals;
    double convergence;
};

struct PIparam *pi_param_create(double convergence, struct ValueF *policy)
{ /* bellman.c:1446-1464 */
    struct PIparam *p = xcalloc(1, sizeof(*p));
    p->convergence = convergence;
    p->vf_policy = policy;
    return p;
}
void pi_param_destroy(struct PIparam *p) { free(p); }
void pi_param_add_cp(struct PIparam *p, struct ControlParams *cp) { p->cp = cp; }
void pi_param_add_value(struct PIparam *p, struct ValueF *vf) { p->vf_iteration = vf; p->niter_evals = 0; p->niter_node_evals = 0; }
size_t pi_param_get_npol_evals(const struct PIparam *p) { return p->npol_evals; }
size_t pi_param_get_niter_node_evals(const struct PIparam *p) { return p->niter_node_evals; }

/* bellman_pi (bellman.c:1702-1886) for F fibers.  Per node the reference caches the policy's [prob, dt, stage] under
 * key2 = (node index, pi_iter); here the cached element is the CANDIDATE INDEX the policy applies (one double), from
 * which the device recomputes the rates.  Counters follow the reference: every node of every call counts as an
 * iteration evaluation (its value memo is never filled -- SURVEY.md 9 Q2; the lookup is kept, the mis-keyed 1-element
 * entries it pushes into the prob table are not), npol_evals counts nodes whose policy had to be computed. */
static double g_tp[4]; /* C3SC_PROFILE: core upload, flags + policy cache lookup, policy pass, evaluation pass */
static size_t g_np;

static int pi_core(struct PIparam *pi, size_t F, size_t k0, const int32_t *idx, const double *x_in, double *out, int fast)
{
    struct ControlParams *cp = pi->cp;
    struct MCAparam *mca = cp->mca;
    struct DPparam *dp = cp->dp;
    const size_t dx = mca->dx, N = mca->ngrid[k0];
    const double t_sync = now_s();
    struct c3sc_hip_ctx *ctx_it = sync_device_ctx(cp, workspace_get_hip_ctx(cp->work), pi->vf_iteration);
    struct c3sc_hip_ctx *ctx_pol = sync_device_ctx(cp, workspace_get_hip_ctx_policy(cp->work), pi->vf_policy);
    const double t_begin = now_s();
    g_tp[0] += t_begin - t_sync;
    g_np++;
    struct HTable *ht_prob = workspace_get_pi_prob_htable(cp->work), *ht_iter = workspace_get_pi_htable(cp->work);
    struct FastMemo *fm = workspace_get_pi_prob_fastmemo(cp->work);
    size_t *ser = workspace_get_ind_to_serialize(cp->work);
    const size_t pi_iter = workspace_get_pi_iter(cp->work), pi_sub = workspace_get_pi_subiter(cp->work);
    char key1[256], key2[256];
    struct FmFiber ff;
    const int have_model = dp_has_device_model(dp);
    const int brute = c3opt_is_bruteforce(cp->opt);
    const size_t pw = brute ? 1 : c3opt_get_d(cp->opt); /* cached policy per node: candidate index, or the control itself */
    double *x_own = NULL;
    const double *x = x_in;
    if (x == NULL) { x_own = fibers_x_from_idx(mca, F, k0, idx); x = x_own; } /* absorbed flags and callbacks need coordinates */

    double *polv = xcalloc(F * N * pw, sizeof(double));
    int *absorbed = xcalloc(N, sizeof(int));
    size_t *nv = xcalloc(2 * N, sizeof(size_t));
    unsigned char *need = xcalloc(F, 1), *stored = xcalloc(F * N, 1), *miss = xcalloc(F * N, 1), *has = xcalloc(F * N, 1);
    size_t fi[C3SC_MAX_DIM], nf[2 * C3SC_MAX_DIM];
    for (size_t f = 0; f < F; f++) {
        for (size_t m = 0; m < dx; m++) { fi[m] = (size_t)idx[f * dx + m]; ser[m] = fi[m]; }
        fi[k0] = 0;
        process_fibers_neighbor(dx, fi, k0, x + f * N * dx, absorbed, nv, nf, mca->ngrid, dp->bound);
        ser[dx] = pi_iter;    /* bellman.c:1759 */
        ser[dx + 1] = pi_sub; /* :1760 */
        if (fast) { fastmemo_fiber_begin(&ff, dx, idx + f * dx, k0, pi_iter, 0); fastmemo_fiber_prefetch(fm, &ff, N); }
        for (size_t j = 0; j < N; j++) {
            if (!fast) {
                ser[k0] = j;
                size_t_a_to_char(ser, dx + 2, key1);
                size_t_a_to_char(ser, dx + 1, key2);
                size_t nb = 0;
                double *v = htable_get_element(ht_iter, key1, &nb); /* :1781 (never filled, Q2) */
                if (v != NULL) { out[f * N + j] = v[0]; stored[f * N + j] = 1; continue; }
            }
            pi->niter_evals++;
            pi->niter_node_evals++;
            if (absorbed[j] == 1 || absorbed[j] == -1) continue; /* :1787, :1794: boundcost / obscost on the device */
            int found = 1;
            if (fast) {
                for (size_t q = 0; q < pw && found; q++) {
                    fastmemo_fiber_counter(&ff, pi_iter, q);
                    found = fastmemo_fiber_get(fm, &ff, j, &polv[(f * N + j) * pw + q]);
                }
            } else {
                size_t nb = 0;
                double *pc = htable_get_element(ht_prob, key2, &nb); /* :1806 */
                found = pc != NULL;
                if (found) memcpy(&polv[(f * N + j) * pw], pc, pw * sizeof(double));
            }
            if (found) has[f * N + j] = 1;
            else { miss[f * N + j] = 1; need[f] = 1; pi->npol_evals++; }
        }
    }
    const size_t U = c3opt_get_nbrute(cp->opt), S = 2 * dx + 1;
    double *tables = NULL, *costs2 = NULL;
    if (!have_model) { /* universal path: host callbacks evaluated once, used by both passes */
        tables = xcalloc(F * N * U * S, sizeof(double));
        costs2 = xcalloc(F * N * 2, sizeof(double));
        for (size_t f = 0; f < F; f++) eval_callback_tables(cp, k0, idx + f * dx, N, x + f * N * dx, tables + f * N * U * S, costs2 + f * N * 2);
    }
    const double t_looked = now_s();
    g_tp[1] += t_looked - t_begin;
    /* policy pass: fibers with a node whose policy is not cached yet -> greedy control for vf_policy (:1832-1846) */
    size_t nrun = 0;
    for (size_t f = 0; f < F; f++) nrun += need[f];
    if (nrun > 0) {
        int32_t *ridx = xcalloc(nrun * dx, sizeof(int32_t)), *rui = xcalloc(nrun * N, sizeof(int32_t));
        double *rout = xcalloc(nrun * N, sizeof(double)), *ruo = xcalloc(nrun * N * pw, sizeof(double));
        size_t r = 0;
        for (size_t f = 0; f < F; f++)
            if (need[f]) memcpy(ridx + (r++) * dx, idx + f * dx, dx * sizeof(int32_t));
        if (!brute) {
            hipok(ctx_pol, c3sc_hip_bellman_fibers_box_host(ctx_pol, (int)k0, nrun, ridx, rout, ruo, NULL), "c3sc_hip_bellman_fibers_box_host");
        } else if (have_model) {
            hipok(ctx_pol, c3sc_hip_bellman_fibers_host(ctx_pol, (int)k0, nrun, ridx, rout, rui, NULL), "c3sc_hip_bellman_fibers_host");
        } else {
            double *rt = xcalloc(nrun * N * U * S, sizeof(double)), *rc2 = xcalloc(nrun * N * 2, sizeof(double));
            r = 0;
            for (size_t f = 0; f < F; f++)
                if (need[f]) {
                    memcpy(rt + r * N * U * S, tables + f * N * U * S, N * U * S * sizeof(double));
                    memcpy(rc2 + r * N * 2, costs2 + f * N * 2, N * 2 * sizeof(double));
                    r++;
                }
            hipok(ctx_pol, c3sc_hip_bellman_fibers_tables_host(ctx_pol, (int)k0, nrun, ridx, rt, rc2, rout, rui, NULL),
                  "c3sc_hip_bellman_fibers_tables_host");
            free(rt); free(rc2);
        }
        if (!fast) die_if_stationary(ctx_pol);
        r = 0;
        for (size_t f = 0; f < F; f++) {
            if (!need[f]) continue;
            for (size_t m = 0; m < dx; m++) ser[m] = (size_t)idx[f * dx + m];
            ser[dx] = pi_iter;
            if (fast) { fastmemo_fiber_begin(&ff, dx, idx + f * dx, k0, pi_iter, 0); fastmemo_fiber_prefetch(fm, &ff, N); }
            for (size_t j = 0; j < N; j++) {
                if (!miss[f * N + j]) continue;
                double *pv = &polv[(f * N + j) * pw];
                has[f * N + j] = 1;
                if (brute) pv[0] = (double)rui[r * N + j];
                else memcpy(pv, &ruo[(r * N + j) * pw], pw * sizeof(double));
                if (fast) {
                    for (size_t q = 0; q < pw; q++) {
                        fastmemo_fiber_counter(&ff, pi_iter, q);
                        fastmemo_fiber_put(fm, &ff, j, pv[q]);
                    }
                } else {
                    ser[k0] = j;
                    size_t_a_to_char(ser, dx + 1, key2);
                    htable_add_element(ht_prob, key2, pv, pw); /* :1877 (there: 2dx+3 doubles) */
                }
            }
            r++;
        }
        free(ridx); free(rui); free(rout); free(ruo);
    }
    const double t_pol = now_s();
    g_tp[2] += t_pol - t_looked;
    /* evaluation pass on vf_iteration with the policy applied (:1807-1815, :1857-1865) */
    double *eout = xcalloc(F * N, sizeof(double));
    int32_t *policy = xcalloc(F * N, sizeof(int32_t));
    for (size_t i = 0; i < F * N; i++) { /* nodes without a policy (absorbed / obstacle): index -1, control 0 */
        if (brute) policy[i] = has[i] ? (int32_t)polv[i] : -1;
        else if (!has[i]) memset(&polv[i * pw], 0, pw * sizeof(double));
    }
    if (!brute) hipok(ctx_it, c3sc_hip_policy_fibers_box_host(ctx_it, (int)k0, F, idx, polv, eout, NULL), "c3sc_hip_policy_fibers_box_host");
    else if (have_model) hipok(ctx_it, c3sc_hip_policy_fibers_host(ctx_it, (int)k0, F, idx, policy, eout, NULL), "c3sc_hip_policy_fibers_host");
    else hipok(ctx_it, c3sc_hip_policy_fibers_tables_host(ctx_it, (int)k0, F, idx, tables, costs2, policy, eout, NULL),
               "c3sc_hip_policy_fibers_tables_host");
    if (!fast) die_if_stationary(ctx_it);
    for (size_t i = 0; i < F * N; i++)
        if (!stored[i]) out[i] = eout[i];
    g_tp[3] += now_s() - t_pol;
    free(eout); free(tables); free(costs2); free(x_own);
    free(policy); free(polv); free(absorbed); free(nv); free(need); free(stored); free(miss); free(has);
    return 0;
}

int bellman_pi_batch(size_t F, size_t N, const double *x, double *out, void *arg)
{
    struct PIparam *pi = arg;
    struct ControlParams *cp = pi->cp;
    assert(cp != NULL && pi->vf_policy != NULL && pi->vf_iteration != NULL);
    struct MCAparam *mca = cp->mca;
    const size_t dx = mca->dx;
    int32_t *idx = xcalloc(F * dx, sizeof(int32_t));
    size_t fi[C3SC_MAX_DIM], k0 = dx;
    for (size_t f = 0; f < F; f++) {
        size_t dv;
        int res = convert_fiber_to_ind(dx, N, x + f * N * dx, mca->ngrid, mca->xgrid, fi, &dv);
        assert(res == 0 && dv < dx && N == mca->ngrid[dv]);
        (void)res;
        if (k0 == dx) k0 = dv;
        if (dv != k0) DIE("bellman_pi_batch: all fibers of a batch must vary the same dimension");
        for (size_t m = 0; m < dx; m++) idx[f * dx + m] = (m == dv) ? 0 : (int32_t)fi[m];
    }
    const int rc = (F > 0) ? pi_core(pi, F, k0, idx, x, out, 0) : 0;
    free(idx);
    return rc;
}

int bellman_pi_batch_idx(size_t F, size_t k, const int32_t *idx, double *out, void *arg)
{ /* new: fibers by grid indices, integer-keyed policy cache (see bellman_vi_batch_idx) */
    struct PIparam *pi = arg;
    assert(pi->cp != NULL && pi->vf_policy != NULL && pi->vf_iteration != NULL && k < pi->cp->mca->dx);
    return (F > 0) ? pi_core(pi, F, k, idx, NULL, out, 1) : 0;
}

int bellman_pi(size_t N, const double *x, double *out, void *arg) { return bellman_pi_batch(1, N, x, out, arg); }

/* =============================================================================== C3Control */
struct C3Control {
    size_t dx, du, dw;
    size_t *ngrid; /* borrowed (bellman.c:1972) */
    double **xgrid, *h, hmin;
    struct Boundary *bound;
    struct MCAparam *mca;
    struct DPparam *dp;
    struct Workspace *work;
    struct ControlParams *cp_active;
    /* implicit policy for simulation (bellman.c:1944-1950, 2034-2042) */
    struct ValueF *policy_sim;
    struct c3Opt *opt_sim;
    void (*transform_sim)(size_t, const double *, double *);
    double *prevpol;
    /* fibers of every core step sharded over the GPUs of a node (c3control_set_fiber_sharding) */
    size_t shard_world, shard_rank;
    c3sc_exchange_fn shard_exchange;
    void *shard_xarg;
    struct c3sc_hip_comm *shard_comm; /* owned: created by c3control_shard_over_gpus */
};

struct C3Control *c3control_create(size_t dx, size_t du, size_t dw, double *lb, double *ub, size_t *ngrid, double discount)
{ /* bellman.c:1962-1999 */
    struct C3Control *c = xcalloc(1, sizeof(*c));
    c->dx = dx; c->du = du; c->dw = dw; c->ngrid = ngrid;
    c->xgrid = xcalloc(dx, sizeof(double *));
    c->h = xcalloc(dx, sizeof(double));
    c->hmin = ub[0] - lb[0];
    size_t maxn = ngrid[0];
    for (size_t m = 0; m < dx; m++) {
        c->xgrid[m] = xcalloc(ngrid[m], sizeof(double));
        for (size_t i = 0; i < ngrid[m]; i++) /* C3 linspace */
            c->xgrid[m][i] = lb[m] + (ub[m] - lb[m]) * (double)i / (double)(ngrid[m] - 1);
        c->h[m] = c->xgrid[m][1] - c->xgrid[m][0];
        if (c->h[m] < c->hmin) c->hmin = c->h[m];
        if (ngrid[m] > maxn) maxn = ngrid[m];
    }
    c->bound = boundary_alloc(dx, lb, ub);
    boundary_set_consistent_ends(c->bound, getenv("C3SC_LITERAL_ENDS") == NULL); /* see c3control_set_consistent_ends */
    c->mca = mca_param_create(dx, du);
    mca_add_grid_refs(c->mca, c->ngrid, c->xgrid, c->hmin, c->h);
    c->dp = dp_param_create(dx, du, dw, discount);
    dp_param_add_boundary(c->dp, c->bound);
    c->work = workspace_alloc(dx, du, dw, maxn);
    return c;
}

void c3control_destroy(struct C3Control *c)
{
    if (c == NULL) return;
    if (c->shard_comm) c3sc_hip_comm_destroy(c->shard_comm); /* before the workspace's device context goes */
    boundary_free(c->bound); mca_param_destroy(c->mca); dp_param_destroy(c->dp); workspace_free(c->work);
    for (size_t m = 0; m < c->dx; m++) free(c->xgrid[m]);
    free(c->prevpol);
    free(c->xgrid); free(c->h); free(c);
}

size_t *c3control_get_ngrid(struct C3Control *c) { return c ? c->ngrid : NULL; }
struct Boundary *c3control_get_boundary(struct C3Control *c) { return c ? c->bound : NULL; }
double **c3control_get_xgrid(struct C3Control *c) { return c ? c->xgrid : NULL; }
void c3control_set_external_boundary(struct C3Control *c, size_t dim, char *type) { boundary_external_set_type(c->bound, dim, type); }
void c3control_add_obstacle(struct C3Control *c, double *center, double *widths) { boundary_add_obstacle(c->bound, center, widths); }
void c3control_add_drift(struct C3Control *c, c3sc_dyn_fn b, void *a) { dp_param_add_drift(c->dp, b, a); }
void c3control_add_diff(struct C3Control *c, c3sc_dyn_fn s, void *a) { dp_param_add_diff(c->dp, s, a); }
void c3control_add_stagecost(struct C3Control *c, int (*f)(double, const double *, const double *, double *, double *)) { dp_param_add_stagecost(c->dp, f); }
void c3control_add_boundcost(struct C3Control *c, int (*f)(double, const double *, double *)) { dp_param_add_boundcost(c->dp, f); }
void c3control_add_obscost(struct C3Control *c, int (*f)(const double *, double *)) { dp_param_add_obscost(c->dp, f); }
void c3control_set_device_model(struct C3Control *c, int model, const double *params, size_t n) { dp_param_set_device_model(c->dp, model, params, n); }

void c3control_set_consistent_ends(struct C3Control *c, int on)
{ /* new, default ON for a C3Control (c3control_create): the solver's fiber function is a function of the node.  The reference's
     own value at the nodes concerned depends on the order in which C3's cross approximation happens to visit them (first memo
     entry wins, bellman.c:1349-1353) -- every value produced here is one the reference can produce.  0 restores
     the literal end-point rule of process_fibers_neighbor (nodeutil.c:570-612). */
    boundary_set_consistent_ends(c->bound, on);
}

void c3control_set_fiber_sharding(struct C3Control *c, size_t world, size_t rank, c3sc_exchange_fn exchange, void *xarg)
{ /* one process per GPU: every rank runs the same solver; step_vi / step_pi evaluate 1/world of each core step's fibers
     on their own device and all-gather the values (world <= 1 or exchange == NULL switches it off) */
    c->shard_world = world;
    c->shard_rank = rank;
    c->shard_exchange = exchange;
    c->shard_xarg = xarg;
}

int c3control_comm_unique_id(void *id128) { return c3sc_hip_comm_unique_id(id128); } /* so that a program links -lc3sc only */

int c3control_shard_over_gpus(struct C3Control *c, size_t world, size_t rank, const void *id128)
{ /* new: one process per GPU of a node, all running the same solver; the fibers of every core step are split over the ranks and
     gathered with one RCCL all-gather (c3sc_hip_comm_*).  id128: the 128 bytes rank 0 got from
     c3sc_hip_comm_unique_id, identical on all ranks.  The device of this rank is C3SC_HIP_DEVICE (default 0: launchers that pin
     one GPU per process, e.g. HIP_VISIBLE_DEVICES, need nothing else).  Collective: every rank must call it.  Value iteration then
     runs its sharded core steps device-resident (c3sc_hip_cross_set_comm); policy iteration goes through the host-driven driver
     with c3sc_hip_comm_exchange as its exchange function.  Returns 0 on success. */
    if (c->shard_comm) { c3sc_hip_comm_destroy(c->shard_comm); c->shard_comm = NULL; }
    c3control_set_fiber_sharding(c, 1, 0, NULL, NULL);
    if (world <= 1 && id128 == NULL) return 0;
    struct c3sc_hip_ctx *ctx = workspace_get_hip_ctx(c->work);
    struct c3sc_hip_comm *comm = NULL;
    const int rc = c3sc_hip_comm_create(ctx, (int)world, (int)rank, id128, &comm);
    if (rc != C3SC_OK) { fprintf(stderr, "c3sc: c3sc_hip_comm_create: %s\n", c3sc_hip_last_error(ctx)); return rc; }
    c->shard_comm = comm;
    c3control_set_fiber_sharding(c, world, rank, c3sc_hip_comm_exchange, comm);
    return 0;
}

struct VIparam *c3control_begin_vi(struct C3Control *c, struct ValueF *vf, struct c3Opt *opt)
{ /* the callback state c3control_step_vi assembles before valuef_interp (bellman.c:2192-2199) */
    c->cp_active = control_params_create(c->dx, c->dw, c->dp, c->mca, c->work, opt);
    struct VIparam *vi = vi_param_create(1e-10);
    vi_param_add_cp(vi, c->cp_active);
    vi_param_add_value(vi, vf);
    workspace_increment_vi_iter(c->work);
    /* entries of earlier iterations can never be hit again (the key carries vi_iter); the reference lets them linger
     * until the next reset, which only costs memory -- and cache misses: keep the integer table small */
    fastmemo_clear(workspace_get_vi_fastmemo(c->work));
    return vi;
}

struct PIparam *c3control_begin_pi(struct C3Control *c, struct ValueF *policy)
{ /* head of c3control_pi_solve (bellman.c:2351-2354): a new policy -> new pi_iter, both tables emptied */
    struct PIparam *pi = pi_param_create(1e-10, policy);
    workspace_increment_pi_iter(c->work);
    workspace_reset_pi_prob_htable(c->work);
    workspace_reset_pi_htable(c->work);
    return pi;
}

void c3control_begin_pi_step(struct C3Control *c, struct PIparam *pi, struct ValueF *vf, struct c3Opt *opt)
{ /* c3control_step_pi before valuef_interp (bellman.c:2236-2249) */
    c->cp_active = control_params_create(c->dx, c->dw, c->dp, c->mca, c->work, opt);
    pi_param_add_cp(pi, c->cp_active);
    pi_param_add_value(pi, vf);
    workspace_increment_pi_subiter(c->work);
}

void c3control_end_pi_step(struct C3Control *c, struct PIparam *pi, size_t *niter_evals)
{
    if (workspace_peek_hip_ctx(c->work)) die_if_stationary(workspace_peek_hip_ctx(c->work));
    if (workspace_peek_hip_ctx_policy(c->work)) die_if_stationary(workspace_peek_hip_ctx_policy(c->work));
    if (getenv("C3SC_PROFILE")) {
        fprintf(stderr, "c3sc profile (policy evaluation): %zu batch calls, core upload %.2f ms, flags + policy cache %.2f ms, policy pass %.2f ms, evaluation pass %.2f ms\n",
                g_np, 1e3 * g_tp[0], 1e3 * g_tp[1], 1e3 * g_tp[2], 1e3 * g_tp[3]);
        g_np = 0; memset(g_tp, 0, sizeof(g_tp));
    }
    if (niter_evals) *niter_evals = pi->niter_node_evals;
    control_params_destroy(c->cp_active);
    c->cp_active = NULL;
    pi_param_add_cp(pi, NULL);
}

void c3control_end_vi(struct C3Control *c, struct VIparam *vi, size_t *nevals)
{
    if (workspace_peek_hip_ctx(c->work)) die_if_stationary(workspace_peek_hip_ctx(c->work));
    if (getenv("C3SC_PROFILE")) {
        fprintf(stderr, "c3sc profile: %zu batch calls, memo lookup %.2f ms, device calls %.2f ms, memo store %.2f ms\n", g_n_calls,
                1e3 * g_t_lookup, 1e3 * g_t_device, 1e3 * g_t_store);
        g_n_calls = 0; g_t_lookup = g_t_device = g_t_store = 0.0;
    }
    if (nevals) *nevals = vi->nnode_evals;
    vi_param_destroy(vi);
    control_params_destroy(c->cp_active);
    c->cp_active = NULL;
}

/* =============================================================================== solver loops (bellman.c:2177-2407) */
struct Diag { /* bellman.c:2409-2420 */
    size_t iter;
    int type; /* 0 policy iteration, 1 value iteration */
    double norm, abs_diff;
    size_t dim;
    size_t *ranks;
    double frac;
    struct Diag *next;
};

void diag_destroy(struct Diag **head)
{
    struct Diag *cur = head ? *head : NULL;
    while (cur != NULL) {
        struct Diag *nx = cur->next;
        free(cur->ranks);
        free(cur);
        cur = nx;
    }
    if (head) *head = NULL;
}

struct Diag *diag_create(size_t iter, int type, double norm, double abs_diff, size_t dim, size_t *ranks, double frac)
{ /* bellman.c:2437-2456: one unlinked record; ranks[0..dim] of the d+1 FT ranks are kept */
    struct Diag *n = xcalloc(1, sizeof(*n));
    n->iter = iter; n->type = type; n->norm = norm; n->abs_diff = abs_diff; n->dim = dim; n->frac = frac;
    n->ranks = xcalloc(dim + 1, sizeof(size_t));
    memcpy(n->ranks, ranks, (dim + 1) * sizeof(size_t));
    return n;
}

void diag_append(struct Diag **diag, size_t iter, int type, double norm, double abs_diff, size_t dim, size_t *ranks, double frac)
{
    struct Diag *n = diag_create(iter, type, norm, abs_diff, dim, ranks, frac);
    if (*diag == NULL) { *diag = n; return; }
    struct Diag *cur = *diag;
    while (cur->next != NULL) cur = cur->next;
    cur->next = n;
}

void diag_print(struct Diag *head, FILE *fp)
{ /* one line per iteration: iter type norm abs_diff avg_rank max_rank frac (bellman.c:2467-2491) */
    for (struct Diag *c = head; c != NULL; c = c->next) {
        double avg = 0.0;
        size_t mx = 0;
        for (size_t i = 1; i < c->dim; i++) { if (c->ranks[i] > mx) mx = c->ranks[i]; avg += (double)c->ranks[i]; }
        avg /= (double)(c->dim - 1);
        fprintf(fp, "%zu %d %3.15G %3.15G %3.15G %zu %3.15G \n", c->iter, c->type, c->norm, c->abs_diff, avg, mx, c->frac);
    }
}

int diag_save(struct Diag *head, char *filename)
{
    FILE *fp = fopen(filename, "w");
    if (fp == NULL) { fprintf(stderr, "cat: can't open %s\n", filename); return 1; }
    diag_print(head, fp);
    fclose(fp);
    return 0;
}

size_t diag_count(const struct Diag *head) { size_t n = 0; for (; head; head = head->next) n++; return n; }
double diag_last_diff(const struct Diag *head) { double d = 0.0; for (; head; head = head->next) d = head->abs_diff; return d; }

struct ValueF *c3control_init_value(struct C3Control *c, int (*f)(size_t, const double *, double *, void *), void *args,
                                    struct ApproxArgs *aargs, int verbose)
{ /* bellman.c:2264-2280 */
    return valuef_interp(c->dx, f, args, c->ngrid, c->xgrid, NULL, aargs, verbose);
}

/* The compiled kernels end at some FT rank per (model, dimension); the reference has no such limit, so a user's
 * maxrank above it is clamped for the interpolation (a copy: the caller's ApproxArgs is borrowed) with one warning,
 * instead of dying in the middle of a solve when the rank adaptation gets there. */
static struct ApproxArgs *device_rank_cap(struct C3Control *c, struct ApproxArgs *in)
{
    struct ApproxArgs *a = approx_args_init();
    approx_args_set_function_class(a, approx_args_get_function_class(in));
    approx_args_set_cross_tol(a, approx_args_get_cross_tol(in));
    approx_args_set_round_tol(a, approx_args_get_round_tol(in));
    approx_args_set_kickrank(a, approx_args_get_kickrank(in));
    approx_args_set_startrank(a, approx_args_get_startrank(in));
    approx_args_set_maxrank(a, approx_args_get_maxrank(in));
    approx_args_set_adapt(a, approx_args_get_adapt(in));
    approx_args_set_crossrank(a, approx_args_get_crossrank(in));
    approx_args_set_cross_maxiter(a, approx_args_get_cross_maxiter(in));
    const int cap = c3sc_hip_max_rank(dp_has_device_model(c->dp) ? c->dp->model : C3SC_MODEL_TABLE, (int)c->dx);
    if (cap > 0 && approx_args_get_maxrank(a) > (size_t)cap) {
        static int warned = 0;
        size_t minN = c->ngrid[0];
        for (size_t m = 0; m < c->dx; m++) if (c->ngrid[m] < minN) minN = c->ngrid[m];
        if (!warned && minN > (size_t)cap) {
            fprintf(stderr, "c3sc: maxrank %zu exceeds the largest rank compiled for this model and dimension (%d); using %d\n",
                    approx_args_get_maxrank(a), cap, cap);
            warned = 1;
        }
        approx_args_set_maxrank(a, (size_t)cap);
    }
    return a;
}

/* Do the core steps of this interpolation fit the device's one-workgroup factorisation (c3sc_hip_cross_setup: at most 8192 rows,
 * N r, of a fiber matrix; ranks up to 48)?  A grid of 200 nodes at cross rank 48 does not: the host-driven cross (same results,
 * one launch per core step) serves it instead of ending the solve in cross_setup. */
static int device_cross_fits(const struct C3Control *c, struct ApproxArgs *aa)
{
    size_t r = approx_args_get_maxrank(aa);
    const size_t xr = approx_args_get_crossrank(aa);
    const char *e = getenv("C3SC_CROSS_RANK_FACTOR");
    if (xr > r) r = xr;
    if (e && atof(e) > 1.0) r = (size_t)ceil(atof(e) * (double)approx_args_get_maxrank(aa));
    if (r > 48) r = 48;
    for (size_t m = 0; m < c->dx; m++)
        if (r * c->ngrid[m] > 8192) {
            static int told = 0;
            if (!told) { fprintf(stderr, "c3sc: %zu nodes at rank %zu exceed the device's core step (8192 rows): the cross iteration is driven from the host\n", c->ngrid[m], r); told = 1; }
            return 0;
        }
    return 1;
}

struct ValueF *c3control_step_vi(struct C3Control *c, struct ValueF *vf, struct ApproxArgs *apargs, struct c3Opt *opt, int verbose,
                                 size_t *nevals)
{ /* bellman.c:2177-2212; the interpolation asks for whole core steps, each one kernel launch */
    const int prof = getenv("C3SC_PROFILE") != NULL;
    const double t_step = prof ? now_s() : 0.0;
    double t_sync = 0.0, t_interp = 0.0;
    struct VIparam *vi = c3control_begin_vi(c, vf, opt);
    struct ApproxArgs *aa = device_rank_cap(c, apargs);
    struct ValueF *next;
    /* Whole cross iterations on the device (c3sc_hip_cross_*: index lists, Bellman launches, node memo and factorisations of all
     * core steps on one stream) when the fibers are the device model's; the host-driven path below serves host callbacks
     * (tables), sharded runs, the very first sweep while the device model is still to be cross-checked against the user's
     * callbacks (vi_core does that on its first fiber), and C3SC_HOST_CROSS=1.  Same results either way. */
    const int rccl = c->shard_comm != NULL && c->shard_exchange == c3sc_hip_comm_exchange; /* sharded over the C communicator */
    const int sharded = (c->shard_world > 1 && c->shard_exchange != NULL) && !rccl;
    const int checked = c->dp->model_checked || c->dp->stagecost == NULL || c->dp->boundcost == NULL || c->dp->obscost == NULL;
    if (dp_has_device_model(c->dp) && !sharded && checked && getenv("C3SC_HOST_CROSS") == NULL && device_cross_fits(c, aa)) {
        const double t0 = prof ? now_s() : 0.0;
        struct c3sc_hip_ctx *ctx = sync_device(vi);
        c3sc_hip_cross_set_comm(ctx, rccl ? c->shard_comm : NULL);
        size_t nodes = 0;
        const double t1 = prof ? now_s() : 0.0;
        next = c3sc_interp_device(c->dx, ctx, !c3opt_is_bruteforce(opt), c->ngrid, c->xgrid, vf, aa, verbose, &nodes, NULL, 0, NULL);
        if (prof) { t_sync = t1 - t0; t_interp = now_s() - t1; }
        vi->nnode_evals += nodes;
        vi->nstate_evals += nodes;
    } else
        next = c3sc_interp_idx_sharded(c->dx, bellman_vi_batch_idx, vi, c->ngrid, c->xgrid, vf, aa, verbose, c->shard_world,
                                       c->shard_rank, c->shard_exchange, c->shard_xarg, vi_absorb_foreign);
    approx_args_free(aa);
    const double t_end0 = prof ? now_s() : 0.0;
    c3control_end_vi(c, vi, nevals);
    if (prof)
        fprintf(stderr, "c3sc step_vi profile: total %.3f ms = begin %.3f, value upload (sync_device) %.3f, interpolation %.3f, end (status read) %.3f\n",
                1e3 * (now_s() - t_step), 1e3 * (t_end0 - t_step - t_sync - t_interp), 1e3 * t_sync, 1e3 * t_interp, 1e3 * (now_s() - t_end0));
    return next;
}

struct ValueF *c3control_step_pi(struct C3Control *c, struct ValueF *vf, struct PIparam *poli, struct ApproxArgs *apargs,
                                 struct c3Opt *opt, int verbose, size_t *niter_evals)
{ /* bellman.c:2214-2262 */
    c3control_begin_pi_step(c, poli, vf, opt);
    struct ApproxArgs *aa = device_rank_cap(c, apargs);
    struct ValueF *next;
    /* whole cross iterations on the device, as in c3control_step_vi: candidate lists of a device model, unsharded */
    const int sharded = c->shard_world > 1 && c->shard_exchange != NULL;
    const int checked = c->dp->model_checked || c->dp->stagecost == NULL || c->dp->boundcost == NULL || c->dp->obscost == NULL;
    if (dp_has_device_model(c->dp) && c3opt_is_bruteforce(opt) && !sharded && checked && getenv("C3SC_HOST_CROSS") == NULL && device_cross_fits(c, aa)) {
        struct ControlParams *cp = poli->cp;
        struct c3sc_hip_ctx *ctx_it = sync_device_ctx(cp, workspace_get_hip_ctx(cp->work), poli->vf_iteration);
        struct c3sc_hip_ctx *ctx_pol = sync_device_ctx(cp, workspace_get_hip_ctx_policy(cp->work), poli->vf_policy);
        size_t npol = 0, requested = 0;
        c3sc_hip_cross_set_comm(ctx_it, NULL); /* a one-rank communicator left by c3control_step_vi: nothing to exchange here */
        next = c3sc_interp_device(c->dx, ctx_it, 0, c->ngrid, c->xgrid, vf, aa, verbose, &npol, ctx_pol, (long long)workspace_get_pi_iter(cp->work),
                                  &requested);
        poli->npol_evals += npol;
        poli->niter_evals += requested;
        poli->niter_node_evals += requested;
    } else
        next = valuef_interp_idx_sharded(c->dx, bellman_pi_batch_idx, poli, c->ngrid, c->xgrid, vf, aa, verbose, c->shard_world,
                                         c->shard_rank, c->shard_exchange, c->shard_xarg);
    approx_args_free(aa);
    c3control_end_pi_step(c, poli, niter_evals);
    return next;
}

static void report(const char *what, size_t ii, size_t maxiter, double diff, double norm, double frac)
{
    printf("\t %s (%zu\\%zu):\n", what, ii + 1, maxiter);
    printf("\t \t L2 Difference between iterates    = %3.5E\n ", diff);
    printf("\t \t L2 Norm of current value function = %3.5E\n", norm);
    printf("\t \t Relative L2 Cauchy difference     = %3.5E\n", diff / norm);
    printf("\t \t Fraction of states evaluated      = %3.5E\n", frac);
}

struct ValueF *c3control_vi_solve(struct C3Control *c, size_t maxiter, double abs_conv_tol, struct ValueF *vo,
                                  struct ApproxArgs *apargs, struct c3Opt *opt, int verbose, struct Diag **diag)
{ /* bellman.c:2282-2340 */
    struct ValueF *start = valuef_copy(vo);
    workspace_reset_vi_htable(c->work);
    double stot = 1.0;
    for (size_t m = 0; m < c->dx; m++) stot *= (double)c->ngrid[m];
    for (size_t ii = 0; ii < maxiter; ii++) {
        if (ii % 1000 == 0) workspace_reset_vi_htable(c->work); /* precaution against memory growth (2296-2298) */
        size_t nevals = 0;
        struct ValueF *next = c3control_step_vi(c, start, apargs, opt, verbose - 1, &nevals);
        const double diff = valuef_norm2diff(start, next), norm = valuef_norm(next), frac = (double)nevals / stot;
        if (verbose > 0) report("Value Iteration", ii, maxiter, diff, norm, frac);
        if (diag != NULL) diag_append(diag, ii, 1, norm, diff, c->dx, valuef_get_ranks(next), frac);
        valuef_destroy(start);
        start = next;
        if (diff < abs_conv_tol) break;
    }
    return start;
}

struct ValueF *c3control_pi_solve(struct C3Control *c, size_t maxiter, double abs_conv_tol, struct ValueF *policy,
                                  struct ApproxArgs *apargs, struct c3Opt *opt, int verbose, struct Diag **diag)
{ /* bellman.c:2343-2407 */
    struct ValueF *start = valuef_copy(policy);
    struct PIparam *poli = c3control_begin_pi(c, policy);
    double stot = 1.0;
    for (size_t m = 0; m < c->dx; m++) stot *= (double)c->ngrid[m];
    for (size_t ii = 0; ii < maxiter; ii++) {
        size_t nevals = 0;
        struct ValueF *next = c3control_step_pi(c, start, poli, apargs, opt, verbose - 1, &nevals);
        const double diff = valuef_norm2diff(start, next), norm = valuef_norm(next), frac = (double)nevals / stot;
        if (verbose > 0) report("POLICY ITERATION", ii, maxiter, diff, norm, frac);
        if (diag != NULL) diag_append(diag, ii, 0, norm, diff, c->dx, valuef_get_ranks(next), frac);
        valuef_destroy(start);
        start = next;
        if (diff < abs_conv_tol) break;
    }
    pi_param_destroy(poli);
    return start;
}

/* =============================================================================== policy simulation tail */
int mca_get_neighbor_node_costs(size_t d, const double *x, struct Boundary *bound, struct ValueF *vf, const size_t *ngrid,
                                double **xgrid, int *absorbed, double *out)
{ /* nodeutil.c:718-816: value at the 2d neighbours one grid spacing away from an OFF-GRID state; inside an obstacle
     every entry is the value at x; out[2d] is left untouched otherwise */
    if (boundary_in_obstacle(bound, x) == 1) {
        *absorbed = -1;
        const double val = valuef_eval(vf, x);
        for (size_t i = 0; i < 2 * d + 1; i++) out[i] = val;
        return 0;
    }
    *absorbed = 0;
    double *xt = xcalloc(d, sizeof(double));
    memcpy(xt, x, d * sizeof(double));
    for (size_t ii = 0; ii < d; ii++) {
        const double lb = xgrid[ii][0], ub = xgrid[ii][ngrid[ii] - 1], h = xgrid[ii][1] - xgrid[ii][0];
        if (((x[ii] + h) < ub) && (x[ii] - h > lb)) {
            xt[ii] = x[ii] - h; out[2 * ii] = valuef_eval(vf, xt);
            xt[ii] = x[ii] + h; out[2 * ii + 1] = valuef_eval(vf, xt);
        } else if ((x[ii] - h) <= lb) { /* left boundary is hit */
            xt[ii] = x[ii] + h; out[2 * ii + 1] = valuef_eval(vf, xt);
            const enum EBTYPE b = boundary_type_dim(bound, ii, 0);
            if (b == ABSORB || b == REFLECT) xt[ii] = lb;
            else if (b == PERIODIC) xt[ii] = (x[ii] > lb) ? ub - (h - (x[ii] - lb)) : (ub - (lb - x[ii])) - h;
            else { fprintf(stderr, "No boundary specified!\n"); assert(1 == 0); }
            out[2 * ii] = valuef_eval(vf, xt);
        } else { /* right boundary is hit */
            xt[ii] = x[ii] - h; out[2 * ii] = valuef_eval(vf, xt);
            const enum EBTYPE b = boundary_type_dim(bound, ii, 1);
            if (b == ABSORB || b == REFLECT) xt[ii] = ub;
            else if (b == PERIODIC) xt[ii] = (x[ii] < ub) ? lb + (h - (ub - x[ii])) : (lb + (x[ii] - ub)) + h;
            else { fprintf(stderr, "No boundary specified!\n"); assert(1 == 0); }
            out[2 * ii + 1] = valuef_eval(vf, xt);
        }
        xt[ii] = x[ii];
    }
    free(xt);
    return 0;
}

void c3control_add_policy_sim(struct C3Control *c, struct ValueF *pol, struct c3Opt *opt_sim,
                              void (*transform)(size_t, const double *, double *))
{ /* bellman.c:2034-2042: borrows both */
    c->policy_sim = pol;
    c->opt_sim = opt_sim;
    c->transform_sim = transform;
}

int c3control_policy_eval(struct C3Control *c, double t, const double *x, double *u)
{ /* bellman.c:2105-2158: one state, host callbacks (a single node is not GPU work) */
    assert(c->policy_sim != NULL && c->opt_sim != NULL);
    if (c->dp->stagecost == NULL) DIE("c3control_policy_eval: the host callbacks (add_drift/diff/stagecost/...) are needed");
    int *absorbed = workspace_get_absorbed(c->work, 0);
    double *costs = workspace_get_costs(c->work, 0);
    struct ControlParams *cp = control_params_create(c->dx, c->dw, c->dp, c->mca, c->work, c->opt_sim);
    int res = mca_get_neighbor_node_costs(c->dx, x, c->bound, c->policy_sim, c->ngrid, c->xgrid, absorbed, costs);
    assert(res == 0);
    (void)res;
    if (c->prevpol == NULL) c->prevpol = xcalloc(c->du, sizeof(double));
    for (size_t i = 0; i < c->du; i++) u[i] = c->prevpol[i];
    control_params_add_time_and_states(cp, t, 1, x);
    struct Memory mem = {cp, 0};
    double val;
    int res2 = bellman_optimal(c->du, u, &val, &mem);
    assert(res2 == 0);
    (void)res2;
    for (size_t i = 0; i < c->du; i++) c->prevpol[i] = u[i];
    control_params_destroy(cp);
    return 0;
}

int c3control_controller(double t, const double *x, double *u, void *args)
{ /* bellman.c:2160-2175 */
    struct C3Control *c = args;
    if (c->transform_sim == NULL) return c3control_policy_eval(c, t, x, u);
    double *xin = xcalloc(c->dx, sizeof(double));
    c->transform_sim(c->dx, x, xin);
    const int out = c3control_policy_eval(c, t, xin, u);
    free(xin);
    return out;
}

int c3control_simulate(struct C3Control *c, const double *x0, double dt, size_t nsteps, const double *noise, double *traj,
                       double *utraj)
{ /* new: the closed-loop tail the examples run through cdyn (e.g. lqg2d.c:346-383) as plain Euler(-Maruyama):
     x_{n+1} = x_n + b(x_n, u_n) dt + sigma(x_n, u_n) dW_n, u_n = controller(x_n); noise = nsteps x dw standard normals
     or NULL.  traj: (nsteps+1) x dx, utraj: nsteps x du (may be NULL). */
    const size_t dx = c->dx, du = c->du, dw = c->dw;
    double *b = xcalloc(dx, sizeof(double)), *s = xcalloc(dx * dw, sizeof(double)), *u = xcalloc(du, sizeof(double));
    memcpy(traj, x0, dx * sizeof(double));
    const double sq = sqrt(dt);
    for (size_t n = 0; n < nsteps; n++) {
        const double *x = traj + n * dx;
        double *xn = traj + (n + 1) * dx;
        int res = c3control_controller((double)n * dt, x, u, c);
        if (res == 0) res = drift_eval(c->dp->drift, (double)n * dt, x, u, b, NULL);
        if (res == 0 && noise) res = diff_eval(c->dp->diff, (double)n * dt, x, u, s, NULL);
        if (res != 0) { free(b); free(s); free(u); return res; }
        for (size_t i = 0; i < dx; i++) {
            double acc = x[i] + b[i] * dt;
            if (noise)
                for (size_t j = 0; j < dw; j++) acc += s[i * dw + j] * sq * noise[n * dw + j];
            xn[i] = acc;
        }
        if (utraj) memcpy(utraj + n * du, u, du * sizeof(double));
    }
    free(b); free(s); free(u);
    return 0;
}
