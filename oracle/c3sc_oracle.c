/* c3sc_oracle.c -- TEST INFRASTRUCTURE ONLY (see c3sc_oracle.h for the rules and the
 * pinning status).  Scalar C99 restatement of the c3sc Bellman-backup hot path.
 * All file:line citations are relative to /root/reference/.
 */
#ifndef M_PI
#define M_PI 3.14159265358979323846 /* strict ISO C has no M_PI; perch.c uses it (:66, 231) */
#endif
#include "c3sc_oracle.h"

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================
 * small dense helpers (stand in for the CBLAS level-1/2 calls of valuefunc.c:424-578 and
 * bellman.c:95; column-major, unit stride, plain left-to-right accumulation)
 * ==================================================================================== */
static double dot_(size_t n, const double *a, const double *b)
{
    double s = 0.0;
    for (size_t i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* y = A x,  A is nrows x ncols col-major with leading dim nrows */
static void gemv_n(size_t nrows, size_t ncols, const double *A, const double *x, double *y)
{
    for (size_t i = 0; i < nrows; i++) y[i] = 0.0;
    for (size_t j = 0; j < ncols; j++) {
        const double xj = x[j];
        for (size_t i = 0; i < nrows; i++) y[i] += A[i + j * nrows] * xj;
    }
}

/* y = A^T x,  x has nrows entries, y has ncols entries */
static void gemv_t(size_t nrows, size_t ncols, const double *A, const double *x, double *y)
{
    for (size_t j = 0; j < ncols; j++) y[j] = dot_(nrows, A + j * nrows, x);
}

double *orc_linspace(double lb, double ub, size_t N)
{
    /* C3's linspace as used at bellman.c:1977: lb + (ub-lb)*i/(N-1)    */
    double *g = malloc(N * sizeof(double));
    assert(g != NULL);
    if (N == 1) { g[0] = lb; return g; }
    for (size_t i = 0; i < N; i++) g[i] = lb + (ub - lb) * (double)i / (double)(N - 1);
    return g;
}

/* ======================================================================================
 * nodeutil.c:267-406  transition_assemble   (and :82-233 transition_assemble_old)
 * ==================================================================================== */
static int tassemble_core(size_t dx, size_t du, size_t dw, double h2, int hvec_is_t,
                          const double *hvec, const double *drift, const double *grad_drift,
                          const double *ddiff, const double *grad_ddiff, double *prob,
                          double *grad_prob, double *dt, double *grad_dt, double *space, int old_quirk)
{
    if (space != NULL) {
        assert(grad_drift != NULL && grad_ddiff != NULL);
        for (size_t j = 0; j < du; j++) space[j] = 0.0;
    }
    double Q = 0.0;
    int res = 0;
    for (size_t i = 0; i < dx; i++) {
        double t, t2;
        if (hvec_is_t) { /* nodeutil.c:291-292: hvec already holds (h^2/h_i, h^2/h_i^2) */
            t = hvec[2 * i];
            t2 = hvec[2 * i + 1];
        } else { /* nodeutil.c:126-127 */
            t = h2 / hvec[i];
            t2 = t / hvec[i];
        }
        /* only the diagonal of the diffusion is used: nodeutil.c:294 */
        const double sig2 = ddiff[i * dx + i] * ddiff[i * dx + i];
        const double half = t2 * sig2 / 2.0;
        prob[2 * i] = half;
        prob[2 * i + 1] = half;
        /* upwinding with a +-1e-14 dead zone: nodeutil.c:300-305 */
        if (drift[i] < -1e-14) prob[2 * i] -= t * drift[i];
        else if (drift[i] > 1e-14) prob[2 * i + 1] += t * drift[i];
        Q += prob[2 * i];
        Q += prob[2 * i + 1];

        if (grad_prob != NULL) { /* nodeutil.c:311-361 */
            for (size_t j = 0; j < du; j++) {
                const double g = t2 * grad_ddiff[i * dx + i + j * dx * dw];
                grad_prob[2 * i * du + j] = g;
                grad_prob[(2 * i + 1) * du + j] = g;
            }
            if (drift[i] < -1e-14) {
                for (size_t j = 0; j < du; j++) grad_prob[2 * i * du + j] += -t * grad_drift[i + j * dx];
            } else if (drift[i] > 1e-14) {
                for (size_t j = 0; j < du; j++) grad_prob[(2 * i + 1) * du + j] += t * grad_drift[i + j * dx];
            } else {
                for (size_t j = 0; j < du; j++) {
                    if (grad_drift[j * dx + i] < 0) {
                        if (old_quirk) /* nodeutil.c:162: the old routine applies the update twice */
                            for (size_t k = 0; k < du; k++) grad_prob[2 * i * du + k] += -t * grad_drift[i + k * dx];
                        for (size_t k = 0; k < du; k++) grad_prob[2 * i * du + k] -= t * grad_drift[i + k * dx];
                    } else if (grad_drift[j * dx + i] > 0) {
                        for (size_t k = 0; k < du; k++) grad_prob[(2 * i + 1) * du + k] += t * grad_drift[i + k * dx];
                    } else {
                        res = 2;
                    }
                }
            }
            for (size_t j = 0; j < du; j++) space[j] += grad_prob[2 * i * du + j];
            for (size_t j = 0; j < du; j++) space[j] += grad_prob[(2 * i + 1) * du + j];
        }
    }
    if (Q < 1e-14) return 1; /* nodeutil.c:365-367: outputs left untouched */

    *dt = h2 / Q;
    prob[2 * dx] = 1.0;
    if (grad_prob != NULL) { /* nodeutil.c:372-395 */
        const double Q2 = Q * Q;
        const double h2_over = h2 / Q2;
        for (size_t j = 0; j < du; j++) {
            grad_prob[2 * dx * du + j] = 0.0;
            grad_dt[j] = -space[j] * h2_over;
        }
        for (size_t i = 0; i < 2 * dx; i++) {
            for (size_t j = 0; j < du; j++)
                grad_prob[i * du + j] = (Q * grad_prob[i * du + j] - space[j] * prob[i]) / Q2;
            prob[i] /= Q;
            prob[2 * dx] -= prob[i];
            for (size_t j = 0; j < du; j++) grad_prob[2 * dx * du + j] -= grad_prob[i * du + j];
        }
    } else { /* nodeutil.c:396-403 */
        for (size_t i = 0; i < dx; i++) {
            prob[2 * i] /= Q;
            prob[2 * i + 1] /= Q;
            prob[2 * dx] -= prob[2 * i];
            prob[2 * dx] -= prob[2 * i + 1];
        }
    }
    return res;
}

int orc_transition_assemble(size_t dx, size_t du, size_t dw, double h, const double *hvec,
                            const double *drift, const double *grad_drift, const double *ddiff,
                            const double *grad_ddiff, double *prob, double *grad_prob, double *dt,
                            double *grad_dt, double *space)
{
    /* here `h` is already h_min^2 and hvec = (t_i, t2_i) pairs: bellman.c:433,449 pass mca->h2, mca->t */
    return tassemble_core(dx, du, dw, h, 1, hvec, drift, grad_drift, ddiff, grad_ddiff, prob, grad_prob, dt,
                          grad_dt, space, 0);
}

int orc_transition_assemble_old(size_t dx, size_t du, size_t dw, double h, const double *hvec,
                                const double *drift, const double *grad_drift, const double *ddiff,
                                const double *grad_ddiff, double *prob, double *grad_prob, double *dt,
                                double *grad_dt, double *space)
{
    /* nodeutil.c:82-233: h is h_min, hvec the per-dim spacings */
    return tassemble_core(dx, du, dw, h * h, 0, hvec, drift, grad_drift, ddiff, grad_ddiff, prob, grad_prob,
                          dt, grad_dt, space, 1);
}

/* nodeutil.c:408-419 */
size_t orc_convert_x_to_ind(double x, size_t N, const double *grid)
{
    for (size_t i = 0; i < N; i++)
        if (fabs(x - grid[i]) < 1e-14) return i;
    return N;
}

/* nodeutil.c:437-470 */
int orc_convert_fiber_to_ind(size_t d, size_t N, const double *x, const size_t *Ngrid,
                             const double *const *xgrid, size_t *fixed_ind, size_t *dim_vary)
{
    for (size_t i = 0; i < d; i++) {
        fixed_ind[i] = orc_convert_x_to_ind(x[i], Ngrid[i], xgrid[i]);
        if (fixed_ind[i] == Ngrid[i]) return 1;
    }
    *dim_vary = d;
    for (size_t i = 0; i < d; i++) {
        size_t second = orc_convert_x_to_ind(x[i + d], Ngrid[i], xgrid[i]);
        if (second != fixed_ind[i]) { *dim_vary = i; break; }
    }
    if (*dim_vary == d) return 1;
    if (N != Ngrid[*dim_vary]) return 2;
    return 0;
}

/* ======================================================================================
 * boundary.c: external BC per dim (same both sides, :604-614), box obstacles (:246-344,
 * :470-481, :668-680).  At most 10 obstacles (:393).
 * ==================================================================================== */
#define ORC_MAX_OBS 10
struct orc_boundary {
    size_t d;
    enum orc_ebtype *type;
    double *lb, *ub; /* external bounds, informational */
    size_t n;
    double *olb[ORC_MAX_OBS];
    double *oub[ORC_MAX_OBS];
    int consistent_ends; /* not in the reference: see orc_boundary_set_consistent_ends */
};

/* NOT the reference's behaviour (default 0 = literal).  The reference resets the absorbed flag of a fiber's two end points
 * from the varying dimension's own boundary type (nodeutil.c:570-612): a node on an absorbing face of a
 * FIXED dimension, or inside an obstacle, is an ordinary node when it happens to be the end point of a reflecting / periodic
 * fiber and a boundary / obstacle node along every other direction -- its value depends on the direction of the fiber it is
 * computed in, and with the memo on the first direction that reaches it (bellman.c:1349-1353).  With this switch the end
 * points keep the flag every other direction gives them (any dimension on an absorbing face -> 1, else obstacle -> -1), so
 * the fiber function handed to the cross approximation is a function of the node.  Used by the solver-level tests that
 * mirror libc3sc.so's c3control_set_consistent_ends. */
void orc_boundary_set_consistent_ends(struct orc_boundary *b, int on) { b->consistent_ends = on; }

struct orc_boundary *orc_boundary_alloc(size_t d, const double *lb, const double *ub)
{
    struct orc_boundary *b = calloc(1, sizeof(*b));
    assert(b != NULL);
    b->d = d;
    b->type = malloc(d * sizeof(*b->type));
    b->lb = malloc(d * sizeof(double));
    b->ub = malloc(d * sizeof(double));
    for (size_t i = 0; i < d; i++) {
        b->type[i] = ORC_ABSORB; /* boundary.c:387: every dimension starts absorbing */
        b->lb[i] = lb[i];
        b->ub[i] = ub[i];
    }
    b->n = 0;
    return b;
}

void orc_boundary_free(struct orc_boundary *b)
{
    if (b == NULL) return;
    for (size_t i = 0; i < b->n; i++) { free(b->olb[i]); free(b->oub[i]); }
    free(b->type); free(b->lb); free(b->ub); free(b);
}

int orc_boundary_external_set_type(struct orc_boundary *b, size_t dim, const char *type)
{
    if (strcmp(type, "absorb") == 0) b->type[dim] = ORC_ABSORB;
    else if (strcmp(type, "periodic") == 0) b->type[dim] = ORC_PERIODIC;
    else if (strcmp(type, "reflect") == 0) b->type[dim] = ORC_REFLECT;
    else return 1;
    return 0;
}

int orc_boundary_add_obstacle(struct orc_boundary *b, const double *center, const double *lengths)
{
    if (b->n == ORC_MAX_OBS) return 1; /* boundary.c:472-475 exits */
    double *l = malloc(b->d * sizeof(double)), *u = malloc(b->d * sizeof(double));
    for (size_t i = 0; i < b->d; i++) { /* boundary.c:264-267 */
        l[i] = center[i] - lengths[i] / 2.0;
        u[i] = center[i] + lengths[i] / 2.0;
    }
    b->olb[b->n] = l;
    b->oub[b->n] = u;
    b->n++;
    return 0;
}

enum orc_ebtype orc_boundary_type_dim(const struct orc_boundary *b, size_t dim, int right)
{
    (void)right; /* boundary.c:604-614 ignores the side (quirk Q7) */
    return b->type[dim];
}

int orc_boundary_in_obstacle(const struct orc_boundary *b, const double *x)
{
    for (size_t o = 0; o < b->n; o++) { /* boundary.c:329-344, inclusive box */
        int inside = 1;
        for (size_t i = 0; i < b->d; i++)
            if (x[i] < b->olb[o][i] || x[i] > b->oub[o][i]) { inside = 0; break; }
        if (inside) return 1;
    }
    return 0;
}

size_t orc_boundary_get_nobs(const struct orc_boundary *b) { return b->n; }
const double *orc_boundary_obstacle_lb(const struct orc_boundary *b, size_t i) { return b->olb[i]; }
const double *orc_boundary_obstacle_ub(const struct orc_boundary *b, size_t i) { return b->oub[i]; }

/* nodeutil.c:489-627 */
int orc_process_fibers_neighbor(size_t d, const size_t *fixed_ind, size_t dim_vary, const double *x,
                                int *absorbed, size_t *nv, size_t *nf, const size_t *ngrid,
                                const struct orc_boundary *bound)
{
    const size_t N = ngrid[dim_vary];
    for (size_t j = 0; j < N; j++) absorbed[j] = orc_boundary_in_obstacle(bound, x + j * d) ? -1 : 0;

    size_t on = 0;
    for (size_t m = 0; m < d; m++) {
        if (m == dim_vary) continue;
        const size_t i = fixed_ind[m];
        if (i == 0 || i == ngrid[m] - 1) { /* :515-559 (left face is tested first) */
            const int left = (i == 0);
            const enum orc_ebtype b = orc_boundary_type_dim(bound, m, left ? 0 : 1);
            if (b == ORC_ABSORB) {
                nf[on] = i;
                nf[on + 1] = i;
                for (size_t j = 0; j < N; j++) absorbed[j] = 1; /* whole fiber, overrides obstacle marks */
            } else if (b == ORC_REFLECT) {
                nf[on] = left ? i : i - 1;
                nf[on + 1] = left ? i + 1 : i;
            } else if (b == ORC_PERIODIC) { /* node 0 == node N-1 physically (quirk Q8) */
                nf[on] = left ? ngrid[m] - 2 : i - 1;
                nf[on + 1] = left ? i + 1 : 1;
            } else {
                return 1; /* reference asserts */
            }
        } else {
            nf[on] = i - 1;
            nf[on + 1] = i + 1;
        }
        on += 2;
    }

    /* the two fiber end points are then overwritten by dim_vary's own BC: :570-612 (quirk Q3) */
    enum orc_ebtype b = orc_boundary_type_dim(bound, dim_vary, 0);
    if (b == ORC_ABSORB) { nv[0] = 0; nv[1] = 0; absorbed[0] = 1; }
    else if (b == ORC_REFLECT) { nv[0] = 0; nv[1] = 1; if (!bound->consistent_ends) absorbed[0] = 0; }
    else if (b == ORC_PERIODIC) { nv[0] = N - 2; nv[1] = 1; if (!bound->consistent_ends) absorbed[0] = 0; }
    else return 1;
    b = orc_boundary_type_dim(bound, dim_vary, 1);
    if (b == ORC_ABSORB) { nv[2 * (N - 1)] = N - 1; nv[2 * (N - 1) + 1] = N - 1; absorbed[N - 1] = 1; }
    else if (b == ORC_REFLECT) { nv[2 * (N - 1)] = N - 2; nv[2 * (N - 1) + 1] = N - 1; if (!bound->consistent_ends) absorbed[N - 1] = 0; }
    else if (b == ORC_PERIODIC) { nv[2 * (N - 1)] = N - 2; nv[2 * (N - 1) + 1] = 1; if (!bound->consistent_ends) absorbed[N - 1] = 0; }
    else return 1;

    for (size_t j = 1; j + 1 < N; j++) { /* :615-624 */
        if (absorbed[j] == 0) { nv[2 * j] = j - 1; nv[2 * j + 1] = j + 1; }
        else { nv[2 * j] = j; nv[2 * j + 1] = j; }
    }
    return 0;
}

/* ======================================================================================
 * valuefunc.c: struct ValueF (:62-78) reduced to what the path reads: d, N, ranks, nodal
 * core tables, and the scratch of valuef_eval_fiber_ind_nn (:385-404).
 * ==================================================================================== */
struct orc_valuef {
    size_t d;
    size_t *N;
    size_t *ranks;
    double **cores;
    size_t maxrank, nmax;
    double **fprod, **bprod, *space;
};

struct orc_valuef *orc_valuef_create(size_t d, const size_t *N, const size_t *ranks, const double *const *cores)
{
    struct orc_valuef *vf = calloc(1, sizeof(*vf));
    assert(vf != NULL);
    vf->d = d;
    vf->N = malloc(d * sizeof(size_t));
    vf->ranks = malloc((d + 1) * sizeof(size_t));
    vf->cores = malloc(d * sizeof(double *));
    memcpy(vf->N, N, d * sizeof(size_t));
    memcpy(vf->ranks, ranks, (d + 1) * sizeof(size_t));
    vf->maxrank = 1;
    vf->nmax = 1;
    for (size_t m = 0; m < d; m++) {
        const size_t n = N[m] * ranks[m] * ranks[m + 1];
        vf->cores[m] = malloc(n * sizeof(double));
        memcpy(vf->cores[m], cores[m], n * sizeof(double));
        if (N[m] > vf->nmax) vf->nmax = N[m];
    }
    for (size_t m = 0; m <= d; m++)
        if (ranks[m] > vf->maxrank) vf->maxrank = ranks[m];
    vf->fprod = malloc(d * sizeof(double *));
    vf->bprod = malloc(d * sizeof(double *));
    for (size_t m = 0; m < d; m++) {
        vf->fprod[m] = calloc(vf->maxrank * vf->nmax, sizeof(double));
        vf->bprod[m] = calloc(vf->maxrank * vf->nmax, sizeof(double));
    }
    vf->space = calloc(vf->maxrank, sizeof(double));
    return vf;
}

void orc_valuef_destroy(struct orc_valuef *vf)
{
    if (vf == NULL) return;
    for (size_t m = 0; m < vf->d; m++) { free(vf->cores[m]); free(vf->fprod[m]); free(vf->bprod[m]); }
    free(vf->cores); free(vf->fprod); free(vf->bprod); free(vf->space);
    free(vf->N); free(vf->ranks); free(vf);
}

/* Definition of a nodal FT at a grid multi-index: G_0[i_0] G_1[i_1] ... G_{d-1}[i_{d-1}].
 * (What function_train_eval returns at grid nodes; tprob_test.c:603-904 pins the fiber
 * routine against it to 1e-14.)  Independent of the prefix/suffix code below. */
double orc_valuef_eval_ind(const struct orc_valuef *vf, const size_t *ind)
{
    double *v = malloc(vf->maxrank * sizeof(double)), *w = malloc(vf->maxrank * sizeof(double));
    v[0] = 1.0;
    for (size_t m = 0; m < vf->d; m++) {
        const size_t r0 = vf->ranks[m], r1 = vf->ranks[m + 1];
        const double *G = vf->cores[m] + ind[m] * r0 * r1;
        for (size_t b = 0; b < r1; b++) {
            double s = 0.0;
            for (size_t a = 0; a < r0; a++) s += v[a] * G[a + b * r0];
            w[b] = s;
        }
        double *tmp = v; v = w; w = tmp;
    }
    const double out = v[0];
    free(v); free(w);
    return out;
}

/* valuef_eval off the grid (valuefunc.c:337-343 -> C3 function_train_eval of LINELM cores): every core is the
 * piecewise-linear interpolant of its nodal table, constant continuation outside the grid. */
double orc_valuef_eval(const struct orc_valuef *vf, const double *const *xgrid, const double *x)
{
    double *v = malloc(vf->maxrank * sizeof(double)), *w = malloc(vf->maxrank * sizeof(double));
    v[0] = 1.0;
    for (size_t m = 0; m < vf->d; m++) {
        const size_t N = vf->N[m], r0 = vf->ranks[m], r1 = vf->ranks[m + 1];
        const double *g = xgrid[m];
        size_t i = 0;
        double wt = 0.0;
        if (x[m] <= g[0]) { i = 0; wt = 0.0; }
        else if (x[m] >= g[N - 1]) { i = N - 2; wt = 1.0; }
        else {
            while (i + 2 < N && g[i + 1] <= x[m]) i++;
            wt = (x[m] - g[i]) / (g[i + 1] - g[i]);
        }
        const double *G0 = vf->cores[m] + i * r0 * r1, *G1 = vf->cores[m] + (i + 1) * r0 * r1;
        for (size_t b = 0; b < r1; b++) {
            double s = 0.0;
            for (size_t a = 0; a < r0; a++) s += v[a] * ((1.0 - wt) * G0[a + b * r0] + wt * G1[a + b * r0]);
            w[b] = s;
        }
        double *tmp = v; v = w; w = tmp;
    }
    const double out = v[0];
    free(v); free(w);
    return out;
}

/* mca_get_neighbor_node_costs (nodeutil.c:718-816): stencil of an off-grid state for the implicit policy */
int orc_mca_get_neighbor_node_costs(size_t d, const double *x, const struct orc_boundary *bound, const struct orc_valuef *vf,
                                    const size_t *ngrid, const double *const *xgrid, int *absorbed, double *out)
{
    if (orc_boundary_in_obstacle(bound, x) == 1) { /* :726-733 */
        *absorbed = -1;
        const double val = orc_valuef_eval(vf, xgrid, x);
        for (size_t i = 0; i < 2 * d + 1; i++) out[i] = val;
        return 0;
    }
    *absorbed = 0;
    double xt[32];
    for (size_t i = 0; i < d; i++) xt[i] = x[i];
    for (size_t ii = 0; ii < d; ii++) {
        const double lb = xgrid[ii][0], ub = xgrid[ii][ngrid[ii] - 1], h = xgrid[ii][1] - xgrid[ii][0];
        if (((x[ii] + h) < ub) && (x[ii] - h > lb)) { /* :745 */
            xt[ii] = x[ii] - h; out[2 * ii] = orc_valuef_eval(vf, xgrid, xt);
            xt[ii] = x[ii] + h; out[2 * ii + 1] = orc_valuef_eval(vf, xgrid, xt);
        } else if ((x[ii] - h) <= lb) { /* :752 */
            xt[ii] = x[ii] + h; out[2 * ii + 1] = orc_valuef_eval(vf, xgrid, xt);
            const enum orc_ebtype b = orc_boundary_type_dim(bound, ii, 0);
            if (b == ORC_ABSORB || b == ORC_REFLECT) { xt[ii] = lb; out[2 * ii] = orc_valuef_eval(vf, xgrid, xt); }
            else if (b == ORC_PERIODIC) {
                if (x[ii] > lb) { xt[ii] = ub - (h - (x[ii] - lb)); }
                else { xt[ii] = (ub - (lb - x[ii])) - h; }
                out[2 * ii] = orc_valuef_eval(vf, xgrid, xt);
            } else return 1;
        } else { /* :781 */
            xt[ii] = x[ii] - h; out[2 * ii] = orc_valuef_eval(vf, xgrid, xt);
            const enum orc_ebtype b = orc_boundary_type_dim(bound, ii, 1);
            if (b == ORC_ABSORB || b == ORC_REFLECT) { xt[ii] = ub; out[2 * ii + 1] = orc_valuef_eval(vf, xgrid, xt); }
            else if (b == ORC_PERIODIC) {
                if (x[ii] < ub) { xt[ii] = lb + (h - (ub - x[ii])); }
                else { xt[ii] = (lb + (x[ii] - ub)) + h; }
                out[2 * ii + 1] = orc_valuef_eval(vf, xgrid, xt);
            } else return 1;
        }
        xt[ii] = x[ii];
    }
    return 0;
}

/* valuefunc.c:369-585.  Same prefix (fprod) / suffix (bprod) scheme and the same output
 * layout out[j*(2d+1) + 2m + {0,1}] = (-,+) neighbour in dim m, out[j*(2d+1)+2d] = self. */
int orc_valuef_eval_fiber_ind_nn(struct orc_valuef *vf, const size_t *fixed_ind, size_t dim_vary,
                                 const size_t *neighbors, const size_t *neighbors_vary, double *out)
{
    const size_t d = vf->d, k = dim_vary;
    const size_t *r = vf->ranks;
    double **fprod = vf->fprod, **bprod = vf->bprod, *space = vf->space;
    const size_t nvals = vf->N[k];

    /* prefix row vectors for m < k at the fixed indices (:414-428) */
    for (size_t m = 0; m < k; m++) {
        const double *G = vf->cores[m] + fixed_ind[m] * r[m] * r[m + 1];
        if (m == 0) memcpy(fprod[0], G, r[0] * r[1] * sizeof(double));
        else gemv_t(r[m], r[m + 1], G, fprod[m - 1], fprod[m]);
    }
    /* suffix column vectors for m > k (:432-446) */
    for (size_t m = d - 1; m > k; m--) {
        const double *G = vf->cores[m] + fixed_ind[m] * r[m] * r[m + 1];
        if (m == d - 1) memcpy(bprod[m], G, r[m] * r[m + 1] * sizeof(double));
        else gemv_n(r[m], r[m + 1], G, bprod[m + 1], bprod[m]);
    }
    /* the varying core: per node forward (:453-465) and backward (:468-480) */
    for (size_t j = 0; j < nvals; j++) {
        const double *G = vf->cores[k] + j * r[k] * r[k + 1];
        if (k == 0) memcpy(fprod[k] + j * r[k + 1], G, r[k] * r[k + 1] * sizeof(double));
        else gemv_t(r[k], r[k + 1], G, fprod[k - 1], fprod[k] + j * r[k + 1]);
        if (k == d - 1) memcpy(bprod[k] + j * r[k], G, r[k] * r[k + 1] * sizeof(double));
        else gemv_n(r[k], r[k + 1], G, bprod[k + 1], bprod[k] + j * r[k]);
    }
    /* push the per-node vectors to the far ends (:485-509) */
    for (size_t m = k + 1; m < d; m++) {
        const double *G = vf->cores[m] + fixed_ind[m] * r[m] * r[m + 1];
        for (size_t j = 0; j < nvals; j++) gemv_t(r[m], r[m + 1], G, fprod[m - 1] + j * r[m], fprod[m] + j * r[m + 1]);
    }
    for (size_t mm = k; mm > 0; mm--) {
        const size_t m = mm - 1;
        const double *G = vf->cores[m] + fixed_ind[m] * r[m] * r[m + 1];
        for (size_t j = 0; j < nvals; j++) gemv_n(r[m], r[m + 1], G, bprod[m + 1] + j * r[m + 1], bprod[m] + j * r[m]);
    }

    const size_t S = 2 * d + 1;
    for (size_t j = 0; j < nvals; j++) { /* :514-519 */
        out[j * S + 2 * k] = bprod[0][neighbors_vary[2 * j]];
        out[j * S + 2 * k + 1] = bprod[0][neighbors_vary[2 * j + 1]];
        out[j * S + 2 * d] = bprod[0][j];
    }
    /* neighbours in the dims before k (:522-547) */
    for (size_t m = 0; m < k; m++) {
        for (int s = 0; s < 2; s++) {
            const size_t nb = neighbors[2 * m + s];
            const double *G = vf->cores[m] + nb * r[m] * r[m + 1];
            for (size_t j = 0; j < nvals; j++) {
                if (m == 0) {
                    out[j * S + 2 * m + s] = dot_(r[1], G, bprod[1] + j * r[1]);
                } else {
                    gemv_n(r[m], r[m + 1], G, bprod[m + 1] + j * r[m + 1], space);
                    out[j * S + 2 * m + s] = dot_(r[m], space, fprod[m - 1]);
                }
            }
        }
    }
    /* neighbours in the dims after k (:549-582); index into `neighbors` skips dim k */
    for (size_t m = k + 1; m < d; m++) {
        for (int s = 0; s < 2; s++) {
            const size_t nb = neighbors[2 * (m - 1) + s];
            const double *G = vf->cores[m] + nb * r[m] * r[m + 1];
            for (size_t j = 0; j < nvals; j++) {
                if (m == d - 1) {
                    out[j * S + 2 * m + s] = dot_(r[m], G, fprod[m - 1] + j * r[m]);
                } else {
                    gemv_t(r[m], r[m + 1], G, fprod[m - 1] + j * r[m], space);
                    out[j * S + 2 * m + s] = dot_(r[m + 1], space, bprod[m + 1]);
                }
            }
        }
    }
    return 0;
}

/* nodeutil.c:647-713 */
int orc_mca_get_neighbor_costs(size_t d, size_t N, const double *x, const struct orc_boundary *bound,
                               struct orc_valuef *vf, const size_t *ngrid, const double *const *xgrid,
                               size_t *fixed_ind, size_t *dim_vary, int *absorbed, double *out)
{
    for (size_t j = 0; j < N; j++) {
        absorbed[j] = 0;
        for (size_t s = 0; s < 2 * d + 1; s++) out[j * (2 * d + 1) + s] = 0.0;
    }
    int res = orc_convert_fiber_to_ind(d, N, x, ngrid, xgrid, fixed_ind, dim_vary);
    if (res != 0) return res; /* reference asserts */
    size_t *nv = calloc(2 * N, sizeof(size_t));
    size_t *nf = calloc(2 * (d > 1 ? d - 1 : 1), sizeof(size_t));
    res = orc_process_fibers_neighbor(d, fixed_ind, *dim_vary, x, absorbed, nv, nf, ngrid, bound);
    if (res == 0) res = orc_valuef_eval_fiber_ind_nn(vf, fixed_ind, *dim_vary, nf, nv, out);
    free(nv);
    free(nf);
    return res;
}

/* ======================================================================================
 * bellman.c:88-112 bellmanrhs ; :171-188 mca_add_grid_refs
 * ==================================================================================== */
double orc_bellmanrhs(size_t dx, size_t du, double stage_cost, const double *stage_grad, double discount,
                      const double *prob, const double *prob_grad, double dt, const double *dtgrad,
                      const double *cost, double *grad)
{
    const double ebt = exp(-discount * dt);
    const double ctg = dot_(2 * dx + 1, prob, cost);
    const double out = dt * stage_cost + ebt * ctg;
    if (grad != NULL) {
        for (size_t j = 0; j < du; j++) {
            grad[j] = stage_grad[j] * dt + dtgrad[j] * stage_cost;
            grad[j] += (-discount) * dtgrad[j] * ebt * ctg;
            for (size_t i = 0; i < 2 * dx + 1; i++) grad[j] += ebt * prob_grad[i * du + j] * cost[i];
        }
    }
    return out;
}

void orc_mca_grid_refs(size_t dx, double hmin, const double *hvec, double *h2, double *t)
{
    *h2 = hmin * hmin;
    for (size_t i = 0; i < dx; i++) {
        t[2 * i] = *h2 / hvec[i];
        t[2 * i + 1] = t[2 * i] / hvec[i];
    }
}

/* ======================================================================================
 * hashgrid.c:49-61 size_t_a_to_char ; :75-87 hashchar ; :89-279 chained table
 * ==================================================================================== */
char *orc_size_t_a_to_char(const size_t *arr, size_t n, char *buffer)
{
    int cx = snprintf(buffer, 256, "%zu ", arr[0]);
    for (size_t i = 1; i < n; i++) cx += snprintf(buffer + cx, 256 - cx, "%zu ", arr[i]);
    return buffer;
}

size_t orc_hashchar(size_t size, const char *str)
{
    size_t h = 0;
    for (; *str != '\0'; str++) h = (size_t)*str + (h << 5) - h;
    return h % size;
}

struct orc_hnode {
    char key[256];
    double *data;
    size_t N;
    struct orc_hnode *next;
};
struct orc_htable {
    size_t size, count;
    struct orc_hnode **table;
};

struct orc_htable *orc_htable_create(size_t size)
{
    if (size < 1) return NULL;
    struct orc_htable *ht = malloc(sizeof(*ht));
    ht->size = size;
    ht->count = 0;
    ht->table = calloc(size, sizeof(*ht->table));
    return ht;
}

void orc_htable_destroy(struct orc_htable *ht)
{
    if (ht == NULL) return;
    for (size_t i = 0; i < ht->size; i++) {
        struct orc_hnode *c = ht->table[i];
        while (c != NULL) { struct orc_hnode *n = c->next; free(c->data); free(c); c = n; }
    }
    free(ht->table);
    free(ht);
}

int orc_htable_add_element(struct orc_htable *ht, const char *key, const double *data, size_t N)
{
    /* LIFO push, duplicates are never checked: hashgrid.c:252-261, :100-124 */
    const size_t h = orc_hashchar(ht->size, key);
    struct orc_hnode *n = malloc(sizeof(*n));
    strcpy(n->key, key);
    n->N = N;
    n->data = calloc(N, sizeof(double));
    memcpy(n->data, data, N * sizeof(double));
    n->next = ht->table[h];
    ht->table[h] = n;
    ht->count++;
    return 0;
}

double *orc_htable_get_element(struct orc_htable *ht, const char *key, size_t *N)
{
    *N = 0;
    for (struct orc_hnode *p = ht->table[orc_hashchar(ht->size, key)]; p != NULL; p = p->next)
        if (strcmp(key, p->key) == 0) { *N = p->N; return p->data; }
    return NULL;
}

size_t orc_htable_count(const struct orc_htable *ht) { return ht->count; }

/* ======================================================================================
 * Problem models: restated callbacks of the examples.  params:
 *   DUBINS3D : none.                         dx=3 du=1
 *   SCAR4D   : none.                         dx=4 du=2
 *   CAR7D    : none.                         dx=7 du=2   
 *   LQGND    : {dim, sig_even, sig_odd}      dx=dim du=dim/2
 *   ROSSLER3D: {3, sig, sig_last}            dx=3 du=1   (examples/rossler/rossler.c)
 *   CHAIN    : {dim, sig_first, sig_last, stage_mode}  dx=dim du=1; stage_mode 0 -> 1.0
 *              (double_int.c:126), 1 -> sum_i x_i^2 (synthetic "quad10d")
 * ==================================================================================== */
int orc_model_dims(int model, const double *p, size_t *dx, size_t *du)
{
    switch (model) {
    case ORC_MODEL_DUBINS3D: *dx = 3; *du = 1; return 0;
    case ORC_MODEL_SCAR4D: *dx = 4; *du = 2; return 0;
    case ORC_MODEL_CAR7D: *dx = 7; *du = 2; return 0;
    case ORC_MODEL_LQGND: *dx = (size_t)p[0]; *du = (size_t)p[0] / 2; return 0;
    case ORC_MODEL_CHAIN: *dx = (size_t)p[0]; *du = 1; return 0;
    case ORC_MODEL_ROSSLER3D: *dx = 3; *du = 1; return 0;
    case ORC_MODEL_TPROB3D: *dx = 3; *du = 3; return 0;
    case ORC_MODEL_PERCH7D: *dx = 7; *du = 1; return 0;
    case ORC_MODEL_SKID5D: *dx = 5; *du = 1; return 0;
    case ORC_MODEL_COTHRUST6D: *dx = 6; *du = 3; return 0;
    default: return 1;
    }
}

int orc_model_drift(int model, const double *p, const double *x, const double *u, double *out)
{
    switch (model) {
    case ORC_MODEL_DUBINS3D: /* dubinscar.c:48-50 */
        out[0] = cos(x[2]);
        out[1] = sin(x[2]);
        out[2] = u[0];
        return 0;
    case ORC_MODEL_SCAR4D: { /* scar.c:58-72 with order = {0,1,2,3} */
        const double orient = x[2], speed = x[3];
        const double L = 0.2, vc = 8.0, alpha = 2.0;
        const double pre = (1.0 / (1.0 + (speed / vc))) * (speed / L);
        out[0] = speed * cos(orient);
        out[1] = speed * sin(orient);
        out[2] = pre * tan(u[0]);
        out[3] = alpha * u[1];
        return 0;
    }
    case ORC_MODEL_CAR7D: { /* state (x,y,theta,v,omega,delta,a) */
        const double th = x[2], v = x[3], om = x[4], de = x[5], a = x[6];
        out[0] = v * cos(th);
        out[1] = v * sin(th);
        out[2] = om;
        out[3] = 2.0 * a;
        out[4] = (v / (0.2 * (1.0 + v / 8.0)) * tan(de) - om) / 0.5;
        out[5] = u[0];
        out[6] = u[1];
        return 0;
    }
    case ORC_MODEL_LQGND: { /* lqgnd.c:86-96 */
        const size_t dim = (size_t)p[0];
        size_t on = 0;
        for (size_t i = 0; i < dim; i++) {
            if ((i % 2) == 0) out[i] = x[i + 1];
            else out[i] = u[on++];
        }
        return 0;
    }
    case ORC_MODEL_CHAIN: { /* double_int.c:86-89 */
        const size_t dim = (size_t)p[0];
        for (size_t i = 0; i + 1 < dim; i++) out[i] = x[i + 1];
        out[dim - 1] = u[0];
        return 0;
    }
    case ORC_MODEL_PERCH7D: { /* perch.c:36-139, the reference's own expressions (libm cos / sin / atan2) */
        const double m = 0.05, g = 9.81, rho = 1.292, S_w = 0.1, S_e = 0.025, In = 6e-3, l = 0.35, l_w = -0.03, l_e = 0.04;
        const double c_t = cos(x[2]), c_tp = cos(x[2] + x[3]), c_p = cos(x[3]), s_t = sin(x[2]), s_tp = sin(x[2] + x[3]);
        const double dx_w[2] = {x[4] + l_w * x[6] * s_t, x[5] - l_w * x[6] * c_t};
        const double dx_w_norm_sq = dx_w[0] * dx_w[0] + dx_w[1] * dx_w[1];
        const double dx_e[2] = {x[4] + l * x[6] * s_t + l_e * (x[6] + u[0]) * s_tp, x[5] - l * x[6] * c_t - l_e * (x[6] + u[0]) * c_tp};
        const double dx_e_norm_sq = dx_e[0] * dx_e[0] + dx_e[1] * dx_e[1];
        const double alpha_w = x[2] - atan2(dx_w[1], dx_w[0]);
        const double alpha_e = x[2] + x[3] - atan2(dx_e[1], dx_e[0]);
        const double f_w = rho * S_w * dx_w_norm_sq * sin(alpha_w);
        const double f_e = rho * S_e * dx_e_norm_sq * sin(alpha_e);
        out[0] = x[4];
        out[1] = x[5];
        out[2] = x[6];
        out[3] = u[0];
        out[4] = (-f_w * s_t - f_e * s_tp) / m;
        out[5] = (f_w * c_t + f_e * c_tp - m * g) / m;
        out[6] = (-f_w * l_w - f_e * (l * c_p + l_e)) / In;
        return 0;
    }
    case ORC_MODEL_SKID5D: { /* skidding5d/scar.c:39-109 with order = {0,1,2,3,4} */
        const double orient = x[2], angvel = x[3], speed = x[4], steering = u[0];
        const double m = 1460.0, cf = 17000.0, ct = 20000.0, a = 1.2, b = 1.5, In = 2170.0, s = 27.0;
        const double co = cos(orient), so = sin(orient);
        const double ff = cf * ((speed + a * angvel) / s + steering);
        const double ft = ct * (speed - b * angvel) / s;
        out[0] = s * co - speed * so;
        out[1] = s * so + speed * co;
        out[2] = angvel;
        out[3] = (a * ff - b * ft) / In;
        out[4] = -s * angvel + (ff + ft) / m;
        return 0;
    }
    case ORC_MODEL_COTHRUST6D: { /* copterposethrust.c:40-117 with order = {0,..,5} (the callback's range checks return 1) */
        const double lbu[3] = {-1.5, -0.4, -0.4}, ubu[3] = {1.5, 0.4, 0.4};
        for (int i = 0; i < 3; i++) if (u[i] < lbu[i] || u[i] > ubu[i]) return 1;
        const double m = 1.227, g = 9.81, mg = m * g;
        const double cphi = cos(u[1]), sphi = sin(u[1]), cth = cos(u[2]), sth = sin(u[2]);
        out[0] = x[3];
        out[1] = x[4];
        out[2] = x[5];
        out[3] = cphi * sth * (u[0] - mg) / m;
        out[4] = -sphi * (u[0] - mg) / m;
        out[5] = g + cth * cphi * (u[0] - mg) / m;
        return 0;
    }
    case ORC_MODEL_TPROB3D: /* tprob_test.c:223-251 (f3) */
        out[0] = x[0] * pow(x[2], 2) * u[0];
        out[1] = -x[1] * u[2] + u[1];
        out[2] = x[0] * x[1] * u[0] + 2 * u[1];
        return 0;
    case ORC_MODEL_ROSSLER3D: { /* rossler.c:89-94: a = b = 0.1, c = 14 */
        const double a = 0.1, b = 0.1, c = 14.0;
        out[0] = -x[1] - x[2];
        out[1] = x[0] + a * x[1] + u[0];
        out[2] = b + x[2] * (x[0] - c);
        return 0;
    }
    default: return 1;
    }
}

int orc_model_diff_diag(int model, const double *p, const double *x, const double *u, double *out)
{
    (void)x; (void)u;
    switch (model) {
    case ORC_MODEL_DUBINS3D: out[0] = 1e0; out[1] = 1e0; out[2] = 1e-2; return 0; /* dubinscar.c:68-72 */
    case ORC_MODEL_SCAR4D: out[0] = 1.0; out[1] = 1.0; out[2] = 1e-2; out[3] = 1e-2; return 0; /* scar.c:105-112 */
    case ORC_MODEL_CAR7D: out[0] = 1.0; out[1] = 1.0; for (int i = 2; i < 7; i++) out[i] = 1e-2; return 0;
    case ORC_MODEL_LQGND: { /* lqgnd.c:135-142 */
        const size_t dim = (size_t)p[0];
        for (size_t i = 0; i < dim; i++) out[i] = ((i % 2) == 0) ? p[1] : p[2];
        return 0;
    }
    case ORC_MODEL_CHAIN: { /* double_int.c:113-117 */
        const size_t dim = (size_t)p[0];
        for (size_t i = 0; i + 1 < dim; i++) out[i] = p[1];
        out[dim - 1] = p[2];
        return 0;
    }
    case ORC_MODEL_ROSSLER3D: out[0] = p[1]; out[1] = p[1]; out[2] = p[2]; return 0; /* rossler.c:113-117 */
    case ORC_MODEL_TPROB3D: out[0] = 1.; out[1] = 1.; out[2] = 1.; return 0; /* tprob_test.c:197-220 (s2) */
    case ORC_MODEL_PERCH7D: for (size_t i = 0; i < 7; i++) out[i] = 1e-9; return 0; /* perch.c:165-176 */
    case ORC_MODEL_SKID5D: { /* skidding5d/scar.c:111-134, written as the callback writes it: the 5 x 5 matrix is zero-filled, then
                                out[0], out[6], out[12], out[28] (sic -- outside the 25 elements; the slab behind it
                                absorbs the store in the reference, a larger buffer does here) and out[24]; transition_assemble reads
                                the diagonal out[m * 5 + m] only (nodeutil.c:294), so the yaw rate's entry out[18] stays 0 */
        double mat[36];
        for (size_t ii = 0; ii < 36; ii++) mat[ii] = 0.0;
        const double vpos = 1e-5, vorient = 1e-5, vspeed = 1e-5;
        mat[0] = vpos; mat[6] = vpos; mat[12] = vorient; mat[28] = vspeed; mat[24] = vspeed;
        for (size_t i = 0; i < 5; i++) out[i] = mat[i * 5 + i];
        return 0;
    }
    case ORC_MODEL_COTHRUST6D: /* copterposethrust.c:128-152 */
        out[0] = 1e-1; out[1] = 1e-1; out[2] = 2e-1; out[3] = 12e-1; out[4] = 12e-1; out[5] = 12e-1;
        return 0;
    default: return 1;
    }
}

int orc_model_stage(int model, const double *p, const double *x, const double *u, double *out)
{
    switch (model) {
    case ORC_MODEL_DUBINS3D: *out = 1.0; return 0;                                   /* dubinscar.c:93 */
    case ORC_MODEL_SCAR4D: *out = 1.0 + pow(x[0], 2) + pow(x[1], 2); return 0;       /* scar.c:131 */
    case ORC_MODEL_CAR7D: *out = 1.0 + pow(x[0], 2) + pow(x[1], 2); return 0;
    case ORC_MODEL_LQGND: { /* lqgnd.c:155-169 */
        const size_t dim = (size_t)p[0];
        double s = 0.0;
        for (size_t i = 0; i < dim; i++) s += x[i] * x[i];
        for (size_t i = 0; i < dim / 2; i++) s += u[i] * u[i];
        *out = s;
        return 0;
    }
    case ORC_MODEL_SKID5D: /* skidding5d/scar.c:136-152 */
        *out = 1.0 + 0.02 * pow(x[0], 2) + 0.02 * pow(x[1], 2);
        *out = *out + pow(x[3], 2) + pow(x[4], 2);
        return 0;
    case ORC_MODEL_COTHRUST6D: /* copterposethrust.c:160-198 */
        *out = 0.0;
        *out = *out + 60.0 + 2 * pow(u[0], 2) + 1 * pow(u[1], 2) + 6 * pow(u[2], 2);
        *out = *out + 8.0 * pow(x[2], 2.0);
        *out = *out + 6.0 * pow(x[1], 2.0);
        *out = *out + 8.0 * pow(x[0], 2.0);
        return 0;
    case ORC_MODEL_PERCH7D: /* perch.c:185-215 */
        *out = 0.0;
        *out += 20.0 * x[0] * x[0];
        *out += 50.0 * x[1] * x[1];
        *out += 10.0 * x[2] * x[2];
        *out += 1.0 * x[3] * x[3];
        *out += 1.0 * x[4] * x[4];
        *out += 1.0 * x[5] * x[5];
        *out += 1.0 * x[6] * x[6];
        *out += 0.1 * u[0] * u[0];
        return 0;
    case ORC_MODEL_TPROB3D: /* tprob_test.c:273-300 (stagecost3d) */
        *out = 0.0;
        *out += 0.2 * x[0] * x[0];
        *out += 0.5 * x[1] * x[1];
        *out += 2.0 * x[2] * x[2];
        *out += 0.1 * u[0] * u[0];
        *out += 0.5 * u[1] * u[1];
        *out += 3.0 * u[2] * u[2];
        return 0;
    case ORC_MODEL_ROSSLER3D: { /* rossler.c:137-143 */
        const double scale = 1e2, rho = 1e0;
        *out = 0.0;
        for (size_t i = 0; i < 3; i++) *out = *out + scale * x[i] * x[i];
        *out += rho * u[0] * u[0];
        return 0;
    }
    case ORC_MODEL_CHAIN: {
        const size_t dim = (size_t)p[0];
        if (p[3] == 0.0) { *out = 1.0; return 0; } /* double_int.c:126 */
        double s = 0.0;
        for (size_t i = 0; i < dim; i++) s += x[i] * x[i];
        *out = s;
        return 0;
    }
    default: return 1;
    }
}

int orc_model_boundcost(int model, const double *p, const double *x, double *out)
{
    (void)p; (void)x;
    switch (model) {
    case ORC_MODEL_DUBINS3D: case ORC_MODEL_SCAR4D: case ORC_MODEL_CAR7D: *out = 10.0; return 0; /* dubinscar.c:108, scar.c:146 */
    case ORC_MODEL_LQGND: *out = 100.0; return 0;  /* lqgnd.c:183 */
    case ORC_MODEL_CHAIN: *out = 1000.0; return 0; /* double_int.c:139 */
    case ORC_MODEL_ROSSLER3D: *out = 1000.0; return 0; /* rossler.c:156 */
    case ORC_MODEL_TPROB3D: *out = 100.0; return 0; /* tprob_test.c:302-309 */
    case ORC_MODEL_SKID5D: /* skidding5d/scar.c:154-163 */
        *out = 0.1 * pow(x[0], 2) + 0.1 * pow(x[1], 2);
        *out = *out + 0.1 * pow(x[3], 2) + 0.1 * pow(x[4], 2);
        return 0;
    case ORC_MODEL_COTHRUST6D: *out = 10.0; return 0; /* copterposethrust.c:200-209 */
    case ORC_MODEL_PERCH7D: /* perch.c:222-241 */
        *out = 0.0;
        *out += 600.0 * x[0] * x[0];
        *out += 400.0 * x[1] * x[1];
        *out += 1.0 / 9.0 * x[2] * x[2];
        *out += 5.0 * (x[2] - M_PI / 2.0) * (x[2] - M_PI / 2.0);
        *out += 1.0 / 9.0 * x[3] * x[3];
        *out += 1.0 * x[4] * x[4];
        *out += 1.0 * (x[5] + 1.5) * (x[5] + 1.5);
        *out += 1.0 / 9.0 * (x[6] + 0.5) * (x[6] + 0.5);
        return 0;
    default: return 1;
    }
}

int orc_model_obscost(int model, const double *p, const double *x, double *out)
{
    (void)model; (void)p; (void)x;
    *out = 0.0; /* every example returns 0: dubinscar.c:120, scar.c:167, lqgnd.c:191, double_int.c:147 */
    return 0;
}

/* ======================================================================================
 * Problem bundle: C3Control (bellman.c:1942-1999) + DPparam (:206-218) + MCAparam
 * (:118-132) + Workspace memo (util.c:758-766) + a brute-force c3Opt.
 * ==================================================================================== */
struct orc_problem {
    size_t dx, du, dw;
    size_t *ngrid;
    double **xgrid;
    double *h, hmin, h2, *t;
    double discount;
    struct orc_boundary *bound;
    /* dynamics: either a built-in model or user callbacks */
    int model;
    double params[8];
    orc_drift_fn b; void *bargs;
    orc_diff_fn s; void *sargs;
    orc_stage_fn stage; orc_bound_fn boundc; orc_obs_fn obsc;
    /* brute-force candidates */
    size_t ncand;
    double *cands;
    /* value function + memo */
    struct orc_valuef *vf;
    struct orc_htable *vi_htable;
    size_t vi_iter;
    size_t nnode_evals;
    /* policy iteration (util.c workspace: pi_iter, pi_subiter, pi_prob_htable, pi_htable) */
    struct orc_htable *pi_prob_htable, *pi_htable;
    size_t pi_iter, pi_subiter;
    size_t npol_evals, niter_evals, niter_node_evals;
};

struct orc_problem *orc_problem_create(size_t dx, size_t du, size_t dw, const double *lb, const double *ub,
                                       const size_t *ngrid, double discount)
{
    struct orc_problem *p = calloc(1, sizeof(*p));
    assert(p != NULL);
    p->dx = dx; p->du = du; p->dw = dw;
    p->ngrid = malloc(dx * sizeof(size_t));
    memcpy(p->ngrid, ngrid, dx * sizeof(size_t));
    p->xgrid = malloc(dx * sizeof(double *));
    p->h = malloc(dx * sizeof(double));
    p->hmin = ub[0] - lb[0]; /* bellman.c:1975 */
    for (size_t i = 0; i < dx; i++) {
        p->xgrid[i] = orc_linspace(lb[i], ub[i], ngrid[i]);
        p->h[i] = p->xgrid[i][1] - p->xgrid[i][0];
        if (p->h[i] < p->hmin) p->hmin = p->h[i];
    }
    p->t = malloc(2 * dx * sizeof(double));
    orc_mca_grid_refs(dx, p->hmin, p->h, &p->h2, p->t);
    p->discount = discount;
    p->bound = orc_boundary_alloc(dx, lb, ub);
    p->vi_htable = orc_htable_create(1000000); /* util.c:758-762 */
    return p;
}

void orc_problem_destroy(struct orc_problem *p)
{
    if (p == NULL) return;
    for (size_t i = 0; i < p->dx; i++) free(p->xgrid[i]);
    free(p->xgrid); free(p->ngrid); free(p->h); free(p->t); free(p->cands);
    orc_boundary_free(p->bound);
    orc_htable_destroy(p->vi_htable);
    orc_htable_destroy(p->pi_prob_htable);
    orc_htable_destroy(p->pi_htable);
    free(p);
}

struct orc_boundary *orc_problem_boundary(struct orc_problem *p) { return p->bound; }
const double *orc_problem_xgrid(const struct orc_problem *p, size_t dim) { return p->xgrid[dim]; }
double orc_problem_h2(const struct orc_problem *p) { return p->h2; }
const double *orc_problem_t(const struct orc_problem *p) { return p->t; }

void orc_problem_set_model(struct orc_problem *p, int model, const double *params, size_t nparams)
{
    p->model = model;
    memset(p->params, 0, sizeof(p->params));
    for (size_t i = 0; i < nparams && i < 8; i++) p->params[i] = params[i];
}

void orc_problem_set_callbacks(struct orc_problem *p, orc_drift_fn b, void *bargs, orc_diff_fn s, void *sargs,
                               orc_stage_fn stage, orc_bound_fn bound, orc_obs_fn obs)
{
    p->model = 0;
    p->b = b; p->bargs = bargs; p->s = s; p->sargs = sargs;
    p->stage = stage; p->boundc = bound; p->obsc = obs;
}

void orc_problem_set_bruteforce(struct orc_problem *p, size_t ncand, const double *cands)
{
    free(p->cands);
    p->ncand = ncand;
    p->cands = malloc(ncand * p->du * sizeof(double));
    memcpy(p->cands, cands, ncand * p->du * sizeof(double));
}

void orc_problem_set_value(struct orc_problem *p, struct orc_valuef *vf) { p->vf = vf; p->nnode_evals = 0; }
void orc_problem_increment_vi_iter(struct orc_problem *p) { p->vi_iter++; }
void orc_problem_reset_vi_htable(struct orc_problem *p)
{
    orc_htable_destroy(p->vi_htable);
    p->vi_htable = orc_htable_create(1000000);
}
size_t orc_problem_nnode_evals(const struct orc_problem *p) { return p->nnode_evals; }

/* bellman_control, no-gradient branch (bellman.c:367-480 with grad_u == NULL) */
static int control_value(struct orc_problem *p, const double *x, const double *u, const double *costs,
                         double *drift, double *diff, double *prob, double *val)
{
    const size_t dx = p->dx, dw = p->dw;
    double stage, dt = 0.0;
    int res;
    if (p->model != 0) {
        res = orc_model_drift(p->model, p->params, x, u, drift);
        if (res) return res;
        for (size_t i = 0; i < dx * dw; i++) diff[i] = 0.0;
        double sd[16];
        res = orc_model_diff_diag(p->model, p->params, x, u, sd);
        if (res) return res;
        for (size_t i = 0; i < dx; i++) diff[i * dx + i] = sd[i];
        res = orc_model_stage(p->model, p->params, x, u, &stage);
        if (res) return res;
    } else {
        res = p->b(0.0, x, u, drift, NULL, p->bargs);
        if (res) return res;
        res = p->s(0.0, x, u, diff, NULL, p->sargs);
        if (res) return res;
        res = p->stage(0.0, x, u, &stage, NULL);
        if (res) return res;
    }
    res = orc_transition_assemble(dx, p->du, dw, p->h2, p->t, drift, NULL, diff, NULL, prob, NULL, &dt, NULL, NULL);
    if (res != 0) return 100 + res; /* bellman.c:452 asserts res == 0 */
    *val = orc_bellmanrhs(dx, p->du, stage, NULL, p->discount, prob, NULL, dt, NULL, costs, NULL);
    return 0;
}

/* bellman_optimal (bellman.c:504-543), BRUTEFORCE branch only.  The scan itself lives in
 * C3 (c3opt_minimize); restated as: candidates in list order, strict '<' (first minimum
 * wins) -- upstream behaviour assumed, parity unpinned. */
static int optimal_value(struct orc_problem *p, int absorbed, const double *x, const double *costs,
                         double *val, int *uidx)
{
    if (absorbed == 1) { /* :513-523 */
        *uidx = -1;
        return p->model ? orc_model_boundcost(p->model, p->params, x, val) : p->boundc(0.0, x, val);
    }
    if (absorbed == -1) { /* :524-532 */
        *uidx = -1;
        return p->model ? orc_model_obscost(p->model, p->params, x, val) : p->obsc(x, val);
    }
    double drift[16], prob[33], diff[256];
    double best = 0.0;
    int bi = -1;
    for (size_t c = 0; c < p->ncand; c++) {
        double v = 0.0;
        int res = control_value(p, x, p->cands + c * p->du, costs, drift, diff, prob, &v);
        if (res) return res;
        if (bi < 0 || v < best) { best = v; bi = (int)c; }
    }
    *val = best;
    *uidx = bi;
    return 0;
}

static void build_fiber_x(const struct orc_problem *p, size_t k, const int *idx, double *x)
{
    const size_t dx = p->dx, N = p->ngrid[k];
    for (size_t j = 0; j < N; j++)
        for (size_t m = 0; m < dx; m++) x[j * dx + m] = (m == k) ? p->xgrid[m][j] : p->xgrid[m][idx[m]];
}

/* ---- policy iteration: c3control_pi_solve head (bellman.c:2351-2354), c3control_step_pi (:2243-2249) ---- */
void orc_problem_pi_begin(struct orc_problem *p)
{
    p->pi_iter++;
    orc_htable_destroy(p->pi_prob_htable);
    p->pi_prob_htable = orc_htable_create(1000000);
    orc_htable_destroy(p->pi_htable);
    p->pi_htable = orc_htable_create(1000000);
    p->npol_evals = 0;
}
void orc_problem_pi_step_begin(struct orc_problem *p)
{
    p->pi_subiter++;
    p->niter_evals = 0;
    p->niter_node_evals = 0;
}
size_t orc_problem_npol_evals(const struct orc_problem *p) { return p->npol_evals; }
size_t orc_problem_niter_node_evals(const struct orc_problem *p) { return p->niter_node_evals; }

/* bellman_optimal at one node, returning the winning candidate (used by bellman_pi to fix the policy) */
static int optimal_control(struct orc_problem *p, const double *x, const double *costs, int *uidx)
{
    double v;
    return optimal_value(p, 0, x, costs, &v, uidx);
}

/* bellman_pi, the live version (bellman.c:1702-1886).  p->vf is vf_iteration, vf_policy the value function the
 * policy is greedy for.  Quirk Q2 is kept: node values are stored in the PROB table under key1 with one element,
 * the value memo (pi_htable) is looked up but never filled, and the final loop keys with keys1[ii]. */
int orc_bellman_pi(struct orc_problem *p, struct orc_valuef *vf_policy, size_t N, const double *x, double *out, int *uidx_out)
{
    const size_t dx = p->dx, S = 2 * dx + 1;
    if (p->pi_prob_htable == NULL) p->pi_prob_htable = orc_htable_create(1000000);
    if (p->pi_htable == NULL) p->pi_htable = orc_htable_create(1000000);
    int *absorbed_pol = calloc(N, sizeof(int)), *absorbed = calloc(N, sizeof(int));
    double *costs_pol = calloc(N * S, sizeof(double)), *costs = calloc(N * S, sizeof(double));
    size_t *fi = calloc(dx, sizeof(size_t));
    size_t dim_vary = 0;
    char (*keys1)[256] = malloc(N * 256), (*keys2)[256] = malloc(N * 256);
    double **probs = calloc(N, sizeof(double *));
    size_t *ind_prob_run = calloc(N, sizeof(size_t));
    size_t nprob_run = 0;
    int res = orc_mca_get_neighbor_costs(dx, N, x, p->bound, vf_policy, p->ngrid, (const double *const *)p->xgrid, fi,
                                         &dim_vary, absorbed_pol, costs_pol); /* :1741 */
    if (res != 0) goto done;
    size_t key_ind[20];
    for (size_t i = 0; i < dx; i++) key_ind[i] = fi[i];
    key_ind[dx] = p->pi_iter;        /* :1759 */
    key_ind[dx + 1] = p->pi_subiter; /* :1760 */
    res = orc_mca_get_neighbor_costs(dx, N, x, p->bound, p->vf, p->ngrid, (const double *const *)p->xgrid, fi, &dim_vary,
                                     absorbed, costs); /* :1767 */
    if (res != 0) goto done;
    for (size_t ii = 0; ii < N; ii++) {
        if (uidx_out) uidx_out[ii] = -1;
        key_ind[dim_vary] = ii;
        orc_size_t_a_to_char(key_ind, dx + 2, keys1[ii]);
        orc_size_t_a_to_char(key_ind, dx + 1, keys2[ii]);
        size_t nb = 0;
        const double *out_stored = orc_htable_get_element(p->pi_htable, keys1[ii], &nb);
        if (out_stored != NULL) {
            out[ii] = out_stored[0];
        } else if (absorbed_pol[ii] == 1) { /* :1787 */
            res = p->model ? orc_model_boundcost(p->model, p->params, x + ii * dx, out + ii) : p->boundc(0.0, x + ii * dx, out + ii);
            if (res) goto done;
            orc_htable_add_element(p->pi_prob_htable, keys1[ii], out + ii, 1);
            p->niter_evals++; p->niter_node_evals++;
        } else if (absorbed[ii] == -1) { /* :1794 */
            res = p->model ? orc_model_obscost(p->model, p->params, x + ii * dx, out + ii) : p->obsc(x + ii * dx, out + ii);
            if (res) goto done;
            orc_htable_add_element(p->pi_prob_htable, keys1[ii], out + ii, 1);
            p->niter_evals++; p->niter_node_evals++;
        } else {
            p->niter_evals++; p->niter_node_evals++;
            const double *cached = orc_htable_get_element(p->pi_prob_htable, keys2[ii], &nb);
            if (cached == NULL) { /* :1808 */
                p->npol_evals++;
                probs[ii] = calloc(S + 2, sizeof(double));
                ind_prob_run[nprob_run++] = ii;
            } else { /* :1814 */
                out[ii] = orc_bellmanrhs(dx, p->du, cached[S + 1], NULL, p->discount, cached, NULL, cached[S], NULL,
                                         costs + ii * S, NULL);
                if (uidx_out) uidx_out[ii] = -2; /* from the cache */
                orc_htable_add_element(p->pi_prob_htable, keys1[ii], out + ii, 1);
            }
        }
    }
    for (size_t r = 0; r < nprob_run; r++) { /* :1832-1869 */
        const size_t ii = ind_prob_run[r];
        int ui = -1;
        res = optimal_control(p, x + ii * dx, costs_pol + ii * S, &ui);
        if (res) goto done;
        if (ui < 0) { res = 3; goto done; }
        const double *u = p->cands + (size_t)ui * p->du;
        double drift[16], diff[256], stage = 0.0;
        if (p->model != 0) {
            double sd[16];
            res = orc_model_drift(p->model, p->params, x + ii * dx, u, drift);
            for (size_t i = 0; i < dx * p->dw; i++) diff[i] = 0.0;
            if (!res) res = orc_model_diff_diag(p->model, p->params, x + ii * dx, u, sd);
            for (size_t i = 0; i < dx; i++) diff[i * dx + i] = sd[i];
            if (!res) res = orc_model_stage(p->model, p->params, x + ii * dx, u, &stage);
        } else {
            res = p->b(0.0, x + ii * dx, u, drift, NULL, p->bargs);
            if (!res) res = p->s(0.0, x + ii * dx, u, diff, NULL, p->sargs);
            if (!res) res = p->stage(0.0, x + ii * dx, u, &stage, NULL);
        }
        if (res) goto done;
        res = orc_transition_assemble(dx, p->du, p->dw, p->h2, p->t, drift, NULL, diff, NULL, probs[ii], NULL,
                                      probs[ii] + S, NULL, NULL);
        if (res) { res = 100 + res; goto done; }
        probs[ii][S + 1] = stage;
        out[ii] = orc_bellmanrhs(dx, p->du, stage, NULL, p->discount, probs[ii], NULL, probs[ii][S], NULL, costs + ii * S, NULL);
        if (uidx_out) uidx_out[ii] = ui;
    }
    for (size_t r = 0; r < nprob_run; r++) { /* :1876-1880 */
        orc_htable_add_element(p->pi_prob_htable, keys2[ind_prob_run[r]], probs[ind_prob_run[r]], S + 2);
        orc_htable_add_element(p->pi_prob_htable, keys1[r], out + ind_prob_run[r], 1);
    }
done:
    for (size_t ii = 0; ii < N; ii++) free(probs[ii]);
    free(probs); free(ind_prob_run); free(keys1); free(keys2);
    free(absorbed_pol); free(absorbed); free(costs_pol); free(costs); free(fi);
    return res;
}

int orc_policy_fibers(struct orc_problem *p, struct orc_valuef *vf_policy, size_t k, size_t F, const int *idx, double *out,
                      int *uidx)
{
    const size_t dx = p->dx, N = p->ngrid[k];
    double *x = malloc(N * dx * sizeof(double));
    int res = 0;
    for (size_t f = 0; f < F && res == 0; f++) {
        build_fiber_x(p, k, idx + f * dx, x);
        res = orc_bellman_pi(p, vf_policy, N, x, out + f * N, uidx ? uidx + f * N : NULL);
    }
    free(x);
    return res;
}

/* c3control_policy_eval (bellman.c:2105-2158): greedy control at an off-grid state; value (out[2d] is not written by
 * mca_get_neighbor_node_costs outside obstacles: the workspace slot keeps whatever it held -- zero here) */
int orc_policy_eval(struct orc_problem *p, const double *x, int *uidx, double *val)
{
    const size_t dx = p->dx, S = 2 * dx + 1;
    double costs[65];
    int absorbed = 0;
    for (size_t i = 0; i < S; i++) costs[i] = 0.0;
    int res = orc_mca_get_neighbor_node_costs(dx, x, p->bound, p->vf, p->ngrid, (const double *const *)p->xgrid, &absorbed, costs);
    if (res) return res;
    return optimal_value(p, absorbed, x, costs, val, uidx);
}

/* bellman.c:1295-1423 */
int orc_bellman_vi(struct orc_problem *p, size_t N, const double *x, double *out, int *uidx, int use_memo)
{
    const size_t dx = p->dx, S = 2 * dx + 1;
    int *absorbed = calloc(N, sizeof(int));
    double *costs = calloc(N * S, sizeof(double));
    size_t *fi = calloc(dx, sizeof(size_t));
    size_t dim_vary = 0;
    /* the FT stencil is evaluated before the memo lookup (quirk Q4, bellman.c:1321 vs 1349) */
    int res = orc_mca_get_neighbor_costs(dx, N, x, p->bound, p->vf, p->ngrid, (const double *const *)p->xgrid, fi,
                                         &dim_vary, absorbed, costs);
    if (res != 0) goto done;

    size_t key_ind[20];
    char key[256];
    for (size_t i = 0; i < dx; i++) key_ind[i] = fi[i];
    key_ind[dx] = 0; /* bellman.c:1337 */
    key_ind[dx + 1] = p->vi_iter;
    for (size_t j = 0; j < N; j++) {
        int ui = -2;
        if (use_memo) {
            key_ind[dim_vary] = j;
            orc_size_t_a_to_char(key_ind, dx + 2, key);
            size_t nb = 0;
            const double *hit = orc_htable_get_element(p->vi_htable, key, &nb);
            if (hit != NULL) {
                out[j] = hit[0];
                if (uidx) uidx[j] = -2;
                continue;
            }
        }
        res = optimal_value(p, absorbed[j], x + j * dx, costs + j * S, &out[j], &ui);
        if (res != 0) goto done;
        if (uidx) uidx[j] = ui;
        p->nnode_evals++;
        if (use_memo) orc_htable_add_element(p->vi_htable, key, &out[j], 1);
    }
done:
    free(absorbed); free(costs); free(fi);
    return res;
}

int orc_bellman_fibers(struct orc_problem *p, size_t k, size_t F, const int *idx, double *out, int *uidx,
                       int *absorbed_out)
{
    const size_t dx = p->dx, N = p->ngrid[k];
    double *x = malloc(N * dx * sizeof(double));
    int res = 0;
    for (size_t f = 0; f < F && res == 0; f++) {
        build_fiber_x(p, k, idx + f * dx, x);
        res = orc_bellman_vi(p, N, x, out + f * N, uidx ? uidx + f * N : NULL, 0);
        if (absorbed_out && res == 0) {
            size_t fi[20], dv, nf[40];
            size_t *nv = malloc(2 * N * sizeof(size_t));
            orc_convert_fiber_to_ind(dx, N, x, p->ngrid, (const double *const *)p->xgrid, fi, &dv);
            orc_process_fibers_neighbor(dx, fi, dv, x, absorbed_out + f * N, nv, nf, p->ngrid, p->bound);
            free(nv);
        }
    }
    free(x);
    return res;
}

int orc_stencil_fibers(struct orc_problem *p, size_t k, size_t F, const int *idx, double *out, int *absorbed_out)
{
    const size_t dx = p->dx, N = p->ngrid[k], S = 2 * dx + 1;
    double *x = malloc(N * dx * sizeof(double));
    int *ab = malloc(N * sizeof(int));
    int res = 0;
    for (size_t f = 0; f < F && res == 0; f++) {
        size_t fi[20], dv;
        build_fiber_x(p, k, idx + f * dx, x);
        res = orc_mca_get_neighbor_costs(dx, N, x, p->bound, p->vf, p->ngrid, (const double *const *)p->xgrid, fi, &dv,
                                         ab, out + f * N * S);
        if (absorbed_out) memcpy(absorbed_out + f * N, ab, N * sizeof(int));
    }
    free(x);
    free(ab);
    return res;
}

/* ---- the two fiber callbacks in the ABI the cross approximation calls (valuefunc.c:615-616:
 * int f(size_t N, const double *x, double *out, void *args)), so that a test can hand them to a cross driver as the
 * black box, the way c3control_step_vi / step_pi do (bellman.c:2201, 2254).  args = struct orc_cb_args. ---- */
int orc_cb_bellman_vi(size_t N, const double *x, double *out, void *args)
{
    struct orc_cb_args *a = args;
    a->ncalls++;
    return orc_bellman_vi(a->p, N, x, out, NULL, a->use_memo);
}

int orc_cb_bellman_pi(size_t N, const double *x, double *out, void *args)
{
    struct orc_cb_args *a = args;
    a->ncalls++;
    return orc_bellman_pi(a->p, a->policy, N, x, out, NULL);
}
