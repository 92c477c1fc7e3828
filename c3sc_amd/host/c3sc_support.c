/* c3sc_support.c -- host-side problem objects of the c3sc C API that the Bellman path reads:
 * Boundary (src/boundary.c), Drift/Diff holders (src/dynamics.c), the string-keyed memo
 * (src/hashgrid.c), ApproxArgs / Workspace (src/util.c) and a brute-force c3Opt.
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

static void *xmalloc(size_t n)
{
    void *p = calloc(1, n ? n : 1);
    if (p == NULL) { fprintf(stderr, "c3sc: out of memory\n"); exit(1); }
    return p;
}

/* ------------------------------------------------------------------------------ Boundary */
#define MAX_OBS 10 /* boundary.c:393 */
struct Box { double *lb, *ub; };
struct Boundary {
    size_t d;
    enum EBTYPE *type;
    double *lo, *hi;
    size_t nobs;
    struct Box obs[MAX_OBS];
    int consistent_ends; /* boundary_set_consistent_ends */
};

struct Boundary *boundary_alloc(size_t d, double *lb, double *ub)
{
    struct Boundary *b = xmalloc(sizeof(*b));
    b->d = d;
    b->type = xmalloc(d * sizeof(*b->type));
    b->lo = xmalloc(d * sizeof(double));
    b->hi = xmalloc(d * sizeof(double));
    for (size_t m = 0; m < d; m++) {
        b->type[m] = ABSORB; /* every dimension starts absorbing (boundary.c:387) */
        b->lo[m] = lb[m];
        b->hi[m] = ub[m];
    }
    b->nobs = 0;
    b->consistent_ends = 0;
    return b;
}

/* NOT in the reference (default 0 = literal): with on = 1, process_fibers_neighbor leaves the flags of a reflecting / periodic
 * fiber's two end points as the fixed dimensions and obstacles set them instead of resetting them (nodeutil.c:570-612, SURVEY.md
 * 9 Q3), so that a node's flag -- and value -- no longer depends on the direction of the fiber it is computed in. */
void boundary_set_consistent_ends(struct Boundary *b, int on) { b->consistent_ends = on ? 1 : 0; }
int boundary_get_consistent_ends(const struct Boundary *b) { return b->consistent_ends; }

struct Boundary *boundary_copy_deep(struct Boundary *o)
{
    if (o == NULL) return NULL;
    struct Boundary *b = boundary_alloc(o->d, o->lo, o->hi);
    memcpy(b->type, o->type, o->d * sizeof(*b->type));
    for (size_t i = 0; i < o->nobs; i++) {
        b->obs[i].lb = xmalloc(o->d * sizeof(double));
        b->obs[i].ub = xmalloc(o->d * sizeof(double));
        memcpy(b->obs[i].lb, o->obs[i].lb, o->d * sizeof(double));
        memcpy(b->obs[i].ub, o->obs[i].ub, o->d * sizeof(double));
    }
    b->nobs = o->nobs;
    b->consistent_ends = o->consistent_ends;
    return b;
}

void boundary_free(struct Boundary *b)
{
    if (b == NULL) return;
    for (size_t i = 0; i < b->nobs; i++) { free(b->obs[i].lb); free(b->obs[i].ub); }
    free(b->type); free(b->lo); free(b->hi); free(b);
}

void boundary_external_set_type(struct Boundary *b, size_t dim, char *type)
{
    assert(b != NULL && dim < b->d);
    if (strcmp(type, "absorb") == 0) b->type[dim] = ABSORB;
    else if (strcmp(type, "periodic") == 0) b->type[dim] = PERIODIC;
    else if (strcmp(type, "reflect") == 0) b->type[dim] = REFLECT;
    else { fprintf(stderr, "Boundary type %s unknown\n", type); exit(1); }
}

void boundary_add_obstacle(struct Boundary *b, double *center, double *lengths)
{
    if (b->nobs == MAX_OBS) { fprintf(stderr, "Not enough space allocated for obstacles in boundary\n"); exit(1); }
    struct Box *o = &b->obs[b->nobs++];
    o->lb = xmalloc(b->d * sizeof(double));
    o->ub = xmalloc(b->d * sizeof(double));
    for (size_t m = 0; m < b->d; m++) { /* boundary.c:264-267 */
        o->lb[m] = center[m] - lengths[m] / 2.0;
        o->ub[m] = center[m] + lengths[m] / 2.0;
    }
}

size_t boundary_get_nobs(struct Boundary *b) { return b->nobs; }
size_t boundary_get_dim(const struct Boundary *b) { return b->d; }
double *boundary_obstacle_get_lb(struct Boundary *b, size_t i) { return b->obs[i].lb; }
double *boundary_obstacle_get_ub(struct Boundary *b, size_t i) { return b->obs[i].ub; }

enum EBTYPE boundary_type_dim(const struct Boundary *b, size_t dim, int right)
{
    (void)right; /* one type per dimension, both faces (boundary.c:604-614) */
    return b->type[dim];
}

double outer_bound_dim(const struct Boundary *b, size_t dim, double x, int *map)
{ /* boundary.c:577-597: a point at or past a periodic face is sent to the opposite face (map 1: left->right, 2: right->left) */
    *map = 0;
    if (x <= b->lo[dim]) {
        if (b->type[dim] == PERIODIC) { *map = 1; return b->hi[dim]; }
    } else if (x >= b->hi[dim]) {
        if (b->type[dim] == PERIODIC) { *map = 2; return b->lo[dim]; }
    }
    return x;
}

/* ---- BoundInfo: where a state sits relative to the faces and obstacles (boundary.c:491-801) */
struct BoundInfo {
    size_t d;
    enum BOUNDRESULT *br;
    enum EBTYPE *type;
    double *xmap;       /* periodic images */
    int absorb_overall; /* on an absorbing face or inside an obstacle */
    int in_obstacle;    /* index of the obstacle, -1 if none */
};

struct BoundInfo *bound_info_alloc(size_t d)
{
    struct BoundInfo *bi = xmalloc(sizeof(*bi));
    bi->d = d;
    bi->br = xmalloc(d * sizeof(*bi->br));
    bi->type = xmalloc(d * sizeof(*bi->type));
    bi->xmap = xmalloc(d * sizeof(double));
    for (size_t i = 0; i < d; i++) { bi->br[i] = IN; bi->type[i] = EB_NONE; bi->xmap[i] = 0.0; }
    bi->absorb_overall = 0;
    bi->in_obstacle = -1;
    return bi;
}
void bound_info_free(struct BoundInfo *bi)
{
    if (bi == NULL) return;
    free(bi->br); free(bi->type); free(bi->xmap); free(bi);
}
int bound_info_set_dim(struct BoundInfo *bi, enum BOUNDRESULT br, enum EBTYPE type, size_t dim)
{ /* boundary.c:550-565: 1 = periodic, the caller still owes the image; -1 = unknown type */
    bi->br[dim] = br;
    bi->type[dim] = type;
    if (type == ABSORB) bi->absorb_overall = 1;
    else if (type == PERIODIC) return 1;
    else if (type != EB_NONE && type != REFLECT) return -1;
    return 0;
}
int bound_info_set_xmap_dim(struct BoundInfo *bi, double x, size_t dim) { bi->xmap[dim] = x; return 0; }
int bound_info_onbound(const struct BoundInfo *bi)
{
    if (bi->in_obstacle > -1) return 1;
    for (size_t i = 0; i < bi->d; i++) if (bi->br[i] != IN) return 1;
    return 0;
}
int bound_info_onbound_dim(const struct BoundInfo *bi, size_t dim) { return bi->in_obstacle > -1 || bi->br[dim] != IN; }
int bound_info_absorb(const struct BoundInfo *bi) { return bi->absorb_overall == 1; }
static int on_face_of_type(const struct BoundInfo *bi, enum EBTYPE t)
{
    for (size_t i = 0; i < bi->d; i++) if (bi->br[i] != IN && bi->type[i] == t) return 1;
    return 0;
}
int bound_info_period(const struct BoundInfo *bi) { return on_face_of_type(bi, PERIODIC); }
int bound_info_reflect(const struct BoundInfo *bi) { return on_face_of_type(bi, REFLECT); }
int bound_info_period_dim_dir(const struct BoundInfo *bi, size_t dim)
{ /* -1 on the left face, 1 otherwise, 0 if the dimension is not periodic (boundary.c:746-757) */
    if (bi->type[dim] != PERIODIC) return 0;
    return bi->br[dim] == LEFT ? -1 : 1;
}
int bound_info_reflect_dim_dir(const struct BoundInfo *bi, size_t dim)
{
    if (bi->type[dim] != REFLECT) return 0;
    return bi->br[dim] == LEFT ? -1 : 1;
}
double bound_info_period_xmap(const struct BoundInfo *bi, size_t dim) { return bi->xmap[dim]; }
int bound_info_get_in_obstacle(const struct BoundInfo *bi) { return bi->in_obstacle; }

static int in_box(const struct Box *bx, size_t d, const double *x)
{
    for (size_t m = 0; m < d; m++) if (x[m] < bx->lb[m] || x[m] > bx->ub[m]) return 0;
    return 1;
}

struct BoundInfo *boundary_type(const struct Boundary *b, double time, const double *x)
{ /* boundary.c:619-662: faces are closed (x <= lo, x >= hi), the first obstacle containing x wins */
    (void)time;
    struct BoundInfo *bi = bound_info_alloc(b->d);
    for (size_t m = 0; m < b->d; m++) {
        if (x[m] <= b->lo[m]) { if (bound_info_set_dim(bi, LEFT, b->type[m], m) == 1) bi->xmap[m] = b->hi[m]; }
        else if (x[m] >= b->hi[m]) { if (bound_info_set_dim(bi, RIGHT, b->type[m], m) == 1) bi->xmap[m] = b->lo[m]; }
    }
    for (size_t i = 0; i < b->nobs; i++)
        if (in_box(&b->obs[i], b->d, x)) { bi->in_obstacle = (int)i; bi->absorb_overall = 1; break; }
    return bi;
}

int boundary_in_obstacle(const struct Boundary *b, const double *x)
{
    for (size_t i = 0; i < b->nobs; i++) {
        size_t m = 0;
        while (m < b->d && !(x[m] < b->obs[i].lb[m] || x[m] > b->obs[i].ub[m])) m++;
        if (m == b->d) return 1; /* inclusive box, boundary.c:329-344 */
    }
    return 0;
}

/* ------------------------------------------------------------------------------ Drift / Diff */
struct Drift { size_t dx, du; c3sc_dyn_fn f; void *args; };
struct Diff { size_t dx, du, dw; c3sc_dyn_fn f; void *args; };

struct Drift *drift_alloc(size_t dx, size_t du)
{
    struct Drift *b = xmalloc(sizeof(*b));
    b->dx = dx; b->du = du;
    return b;
}
void drift_free(struct Drift *b) { free(b); }
void drift_add_func(struct Drift *b, c3sc_dyn_fn f, void *args) { b->f = f; b->args = args; }
size_t drift_get_dx(struct Drift *b) { return b->dx; }
int drift_eval(struct Drift *b, double t, const double *x, const double *u, double *out, double *jac)
{
    if (b->f == NULL) { fprintf(stderr, "Warning: drift dynamics (b->bdyn) not yet specified\n"); return 1; }
    return b->f(t, x, u, out, jac, b->args);
}

struct Diff *diff_alloc(size_t dx, size_t du, size_t dw)
{
    struct Diff *s = xmalloc(sizeof(*s));
    s->dx = dx; s->du = du; s->dw = dw;
    return s;
}
void diff_free(struct Diff *s) { free(s); }
void diff_add_func(struct Diff *s, c3sc_dyn_fn f, void *args) { s->f = f; s->args = args; }
size_t diff_get_dw(struct Diff *s) { return s->dw; }
int diff_eval(struct Diff *s, double t, const double *x, const double *u, double *out, double *grad)
{
    if (s->f == NULL) { fprintf(stderr, "Warning: diff dynamics not yet specified\n"); return 1; }
    return s->f(t, x, u, out, grad, s->args);
}

/* dynamics.c:88-96, 183-192: copies share the callback and its argument pointer */
struct Drift *drift_copy(struct Drift *o)
{
    if (o == NULL) return NULL;
    struct Drift *b = drift_alloc(o->dx, o->du);
    b->f = o->f; b->args = o->args;
    return b;
}
struct Diff *diff_copy(struct Diff *o)
{
    if (o == NULL) return NULL;
    struct Diff *s = diff_alloc(o->dx, o->du, o->dw);
    s->f = o->f; s->args = o->args;
    return s;
}
size_t drift_get_du(struct Drift *b) { return b->du; }

/* dynamics.c:258-354: a (drift, diffusion) pair; dyn_alloc / dyn_init_ref borrow, *_deep own */
struct Dyn { struct Drift *drift; struct Diff *diff; };
struct Dyn *dyn_alloc(struct Drift *b, struct Diff *s)
{
    struct Dyn *d = xmalloc(sizeof(*d));
    d->drift = b; d->diff = s;
    return d;
}
struct Dyn *dyn_copy_deep(struct Dyn *o)
{
    if (o == NULL) return NULL;
    return dyn_alloc(drift_copy(o->drift), diff_copy(o->diff));
}
void dyn_free(struct Dyn *d) { free(d); }
void dyn_free_deep(struct Dyn *d)
{
    if (d == NULL) return;
    drift_free(d->drift);
    diff_free(d->diff);
    free(d);
}
void dyn_init_ref(struct Dyn *d, struct Drift *b, struct Diff *s) { d->drift = b; d->diff = s; }
size_t dyn_get_dx(struct Dyn *d) { return drift_get_dx(d->drift); }
size_t dyn_get_dw(struct Dyn *d) { return diff_get_dw(d->diff); }
size_t dyn_get_du(struct Dyn *d) { return drift_get_du(d->drift); }
int dyn_eval(struct Dyn *d, double t, const double *x, const double *u, double *drift, double *jacdr, double *diff, double *jacdiff)
{ /* dynamics.c:332-349: the diffusion is skipped when the drift callback fails */
    int res = 0;
    if (drift != NULL) res = drift_eval(d->drift, t, x, u, drift, jacdr);
    if (res != 0) return res;
    if (diff != NULL) res = diff_eval(d->diff, t, x, u, diff, jacdiff);
    return res;
}

/* ------------------------------------------------------------------------------ memo (hashgrid.c) */
char *size_t_a_to_char(size_t *arr, size_t n, char *buffer)
{ /* decimal text, one trailing blank per entry, 256-byte buffer (hashgrid.c:49-61) */
    int used = 0;
    for (size_t i = 0; i < n; i++) used += snprintf(buffer + used, (size_t)(256 - used), "%zu ", arr[i]);
    return buffer;
}

size_t c3sc_hashchar(size_t size, const char *s)
{ /* h = c + 31 h with size_t wrap-around (hashgrid.c:75-87) */
    size_t h = 0;
    while (*s != '\0') { h = (size_t)*s + (h << 5) - h; s++; }
    return h % size;
}

struct HNode { struct HNode *next; size_t N; double *data; char key[256]; };
struct HTable { size_t size; struct HNode **bucket; };

struct HTable *htable_create(size_t size)
{
    if (size < 1) return NULL;
    struct HTable *ht = xmalloc(sizeof(*ht));
    ht->size = size;
    ht->bucket = xmalloc(size * sizeof(*ht->bucket));
    return ht;
}

void htable_destroy(struct HTable *ht)
{
    if (ht == NULL) return;
    for (size_t i = 0; i < ht->size; i++)
        for (struct HNode *n = ht->bucket[i]; n != NULL;) { struct HNode *nx = n->next; free(n->data); free(n); n = nx; }
    free(ht->bucket);
    free(ht);
}

int htable_add_element(struct HTable *ht, char *key, double *data, size_t N)
{ /* push front, no duplicate check (hashgrid.c:252-261) */
    struct HNode *n = xmalloc(sizeof(*n));
    strcpy(n->key, key);
    n->N = N;
    n->data = xmalloc(N * sizeof(double));
    memcpy(n->data, data, N * sizeof(double));
    struct HNode **head = &ht->bucket[c3sc_hashchar(ht->size, key)];
    n->next = *head;
    *head = n;
    return 0;
}

double *htable_get_element(struct HTable *ht, char *key, size_t *N)
{
    *N = 0;
    for (struct HNode *n = ht->bucket[c3sc_hashchar(ht->size, key)]; n != NULL; n = n->next)
        if (strcmp(key, n->key) == 0) { *N = n->N; return n->data; }
    return NULL;
}

/* ------------------------------------------------------------------------------ ApproxArgs */
struct ApproxArgs { double cross_tol, round_tol; size_t kickrank, startrank, maxrank, crossrank, cross_maxiter; int adapt; enum function_class fc; };

struct ApproxArgs *approx_args_init(void)
{
    struct ApproxArgs *a = xmalloc(sizeof(*a));
    a->cross_tol = 1e-10; a->round_tol = 1e-10; a->kickrank = 10; a->startrank = 5; a->maxrank = 40; /* util.c:124-130 */
    a->adapt = 1; a->fc = LINELM;
    a->crossrank = 0;
    a->cross_maxiter = 5; /* valuefunc.c:632: ft_cross_args_set_maxiter(fca, 5) */
    return a;
}
void approx_args_free(struct ApproxArgs *a) { free(a); }
void approx_args_set_function_class(struct ApproxArgs *a, enum function_class fc) { a->fc = fc; }
enum function_class approx_args_get_function_class(const struct ApproxArgs *a) { return a->fc; }
void approx_args_set_cross_tol(struct ApproxArgs *a, double v) { a->cross_tol = v; }
double approx_args_get_cross_tol(const struct ApproxArgs *a) { return a->cross_tol; }
void approx_args_set_round_tol(struct ApproxArgs *a, double v) { a->round_tol = v; }
double approx_args_get_round_tol(const struct ApproxArgs *a) { return a->round_tol; }
void approx_args_set_kickrank(struct ApproxArgs *a, size_t v) { a->kickrank = v; }
size_t approx_args_get_kickrank(const struct ApproxArgs *a) { return a->kickrank; }
void approx_args_set_maxrank(struct ApproxArgs *a, size_t v) { a->maxrank = v; }
size_t approx_args_get_maxrank(const struct ApproxArgs *a) { return a->maxrank; }
void approx_args_set_startrank(struct ApproxArgs *a, size_t v) { a->startrank = v; }
size_t approx_args_get_startrank(const struct ApproxArgs *a) { return a->startrank; }
void approx_args_set_adapt(struct ApproxArgs *a, int v) { a->adapt = v; }
/* new: the cross approximation may run at ranks up to `crossrank` (> maxrank) and is then rounded to maxrank by the TT-SVD --
 * an orthogonal projection, so the result is close to the best train of that rank where interpolation through maxrank fibers
 * is 2-4 times further away.  0 = maxrank (the reference's scheme: valuefunc.c:625-649). */
void approx_args_set_crossrank(struct ApproxArgs *a, size_t v) { a->crossrank = v; }
size_t approx_args_get_crossrank(const struct ApproxArgs *a) { return a->crossrank; }
/* new: cap on the cross iterations (left-to-right + right-to-left half sweep) of one interpolation; the reference fixes it at 5
 * (valuefunc.c:632).  Inside a value iteration whose sweeps warm-start from the previous index sets, 1 is enough: the sweeps
 * themselves play the role of the cross iterations (measured on car7d: same step floor, 2.3x fewer core steps per sweep). */
void approx_args_set_cross_maxiter(struct ApproxArgs *a, size_t v) { a->cross_maxiter = v < 1 ? 1 : v; }
size_t approx_args_get_cross_maxiter(const struct ApproxArgs *a) { return a->cross_maxiter; }
int approx_args_get_adapt(const struct ApproxArgs *a) { return a->adapt; }

size_t uniform_stride(size_t N, size_t M)
{ /* largest stride s with s*(M-1) < N-1, at least ... (util.c:995-1006) */
    assert(N >= M);
    size_t s = 1;
    while (s * (M - 1) < (N - 1)) s++;
    return s - 1;
}

/* ------------------------------------------------------------------------------ c3Opt
 * BRUTEFORCE: the candidate list is scanned in order, first strict minimum wins (assumed upstream behaviour).
 * Any other algorithm (the examples ask for BFGS with box bounds): C3's gradient optimiser is not available, the
 * minimiser is a tensor grid over the box + coordinate golden-section polish -- the same algorithm the device runs
 * (kernel_common.hpp: node_backup_box); tolerances and iteration limits are accepted and ignored. */
struct c3Opt {
    enum c3opt_alg alg;
    size_t d, n;
    double *vals;
    double *lb, *ub;
    size_t grid, polish;
    double (*f)(size_t, const double *, double *, void *);
    void *farg;
};

struct c3Opt *c3opt_alloc(enum c3opt_alg alg, size_t d)
{
    struct c3Opt *o = xmalloc(sizeof(*o));
    memset(o, 0, sizeof(*o));
    o->alg = alg; o->d = d;
    o->lb = xmalloc(d * sizeof(double));
    o->ub = xmalloc(d * sizeof(double));
    for (size_t i = 0; i < d; i++) { o->lb[i] = -1.0; o->ub[i] = 1.0; }
    o->grid = d == 1 ? 33 : (d == 2 ? 17 : 9);
    o->polish = 2;
    return o;
}

struct c3Opt *c3opt_copy(struct c3Opt *o)
{
    struct c3Opt *c = xmalloc(sizeof(*c));
    *c = *o;
    c->vals = xmalloc((o->n * o->d + 1) * sizeof(double));
    if (o->n) memcpy(c->vals, o->vals, o->n * o->d * sizeof(double));
    c->lb = xmalloc(o->d * sizeof(double));
    c->ub = xmalloc(o->d * sizeof(double));
    memcpy(c->lb, o->lb, o->d * sizeof(double));
    memcpy(c->ub, o->ub, o->d * sizeof(double));
    return c;
}

void c3opt_free(struct c3Opt *o) { if (o) { free(o->vals); free(o->lb); free(o->ub); free(o); } }

void c3opt_set_brute_force_vals(struct c3Opt *o, size_t n, double *vals)
{
    free(o->vals);
    o->n = n;
    o->vals = xmalloc(n * o->d * sizeof(double));
    memcpy(o->vals, vals, n * o->d * sizeof(double));
}

void c3opt_add_lb(struct c3Opt *o, double *lb) { memcpy(o->lb, lb, o->d * sizeof(double)); }
void c3opt_add_ub(struct c3Opt *o, double *ub) { memcpy(o->ub, ub, o->d * sizeof(double)); }
double *c3opt_get_lb(struct c3Opt *o) { return o->lb; }
double *c3opt_get_ub(struct c3Opt *o) { return o->ub; }
/* knobs of C3's line-search optimisers: accepted so that the examples' set-up code links and runs */
void c3opt_set_relftol(struct c3Opt *o, double v) { (void)o; (void)v; }
void c3opt_set_absxtol(struct c3Opt *o, double v) { (void)o; (void)v; }
void c3opt_set_gtol(struct c3Opt *o, double v) { (void)o; (void)v; }
void c3opt_set_maxiter(struct c3Opt *o, size_t v) { (void)o; (void)v; }
void c3opt_ls_set_maxiter(struct c3Opt *o, size_t v) { (void)o; (void)v; }
void c3opt_ls_set_alpha(struct c3Opt *o, double v) { (void)o; (void)v; }
void c3opt_ls_set_beta(struct c3Opt *o, double v) { (void)o; (void)v; }
void c3opt_set_verbose(struct c3Opt *o, int v) { (void)o; (void)v; }
void c3opt_set_storage_options(struct c3Opt *o, int a, int b, int c) { (void)o; (void)a; (void)b; (void)c; }
/* new: resolution of the box minimiser (grid points per control dimension, polish rounds) */
void c3opt_set_box_search(struct c3Opt *o, size_t grid, size_t polish) { o->grid = grid < 2 ? 2 : grid; o->polish = polish; }
size_t c3opt_get_box_grid(const struct c3Opt *o) { return o->grid; }
size_t c3opt_get_box_polish(const struct c3Opt *o) { return o->polish; }

int c3opt_is_bruteforce(const struct c3Opt *o) { return o->alg == BRUTEFORCE; }
void c3opt_add_objective(struct c3Opt *o, double (*f)(size_t, const double *, double *, void *), void *arg) { o->f = f; o->farg = arg; }
size_t c3opt_get_nbrute(const struct c3Opt *o) { return o->n; }
const double *c3opt_get_brute_vals(const struct c3Opt *o) { return o->vals; }
size_t c3opt_get_d(const struct c3Opt *o) { return o->d; }

static int box_minimize(struct c3Opt *o, double *x, double *val)
{ /* node_backup_box on the host: tensor grid, then coordinate golden section in the best cell */
    const size_t d = o->d, G = o->grid;
    double dl[8], u[8], ub[8];
    size_t total = 1;
    assert(d <= 8);
    for (size_t i = 0; i < d; i++) { dl[i] = (o->ub[i] - o->lb[i]) / (double)(G - 1); total *= G; ub[i] = o->lb[i]; }
    double best = 1.0e301;
    for (size_t c = 0; c < total; c++) {
        size_t rem = c;
        for (size_t i = 0; i < d; i++) { const size_t gi = rem % G; rem /= G; u[i] = (gi == G - 1) ? o->ub[i] : fma((double)gi, dl[i], o->lb[i]); }
        const double v = o->f(d, u, NULL, o->farg);
        if (v < best) { best = v; memcpy(ub, u, d * sizeof(double)); }
    }
    const double gr = 0.6180339887498949;
    for (size_t round = 0; round < o->polish; round++)
        for (size_t i = 0; i < d; i++) {
            double lo = fmax(o->lb[i], ub[i] - dl[i]), hi = fmin(o->ub[i], ub[i] + dl[i]);
            memcpy(u, ub, d * sizeof(double));
            double x1 = hi - gr * (hi - lo), x2 = lo + gr * (hi - lo);
            u[i] = x1; double f1 = o->f(d, u, NULL, o->farg);
            u[i] = x2; double f2 = o->f(d, u, NULL, o->farg);
            for (int it = 0; it < 40; it++) {
                const int left = f1 < f2;
                hi = left ? x2 : hi;
                lo = left ? lo : x1;
                const double xn = left ? hi - gr * (hi - lo) : lo + gr * (hi - lo);
                u[i] = xn;
                const double fn = o->f(d, u, NULL, o->farg);
                const double ox1 = x1, of1 = f1;
                x1 = left ? xn : x2; f1 = left ? fn : f2;
                x2 = left ? ox1 : xn; f2 = left ? of1 : fn;
            }
            const double xm = (f1 < f2) ? x1 : x2, fm = fmin(f1, f2);
            if (fm < best) { best = fm; ub[i] = xm; }
        }
    memcpy(x, ub, d * sizeof(double));
    *val = best;
    return 0;
}

int c3opt_minimize(struct c3Opt *o, double *x, double *val)
{
    assert(o->f != NULL);
    if (o->alg != BRUTEFORCE) return box_minimize(o, x, val);
    /* candidates in list order, first strict minimum wins */
    assert(o->n > 0);
    size_t best = 0;
    double bv = 0.0;
    for (size_t c = 0; c < o->n; c++) {
        double v = o->f(o->d, o->vals + c * o->d, NULL, o->farg);
        if (c == 0 || v < bv) { bv = v; best = c; }
    }
    memcpy(x, o->vals + best * o->d, o->d * sizeof(double));
    *val = bv;
    return 0;
}

/* ------------------------------------------------------------------------------ FastMemo
 * The per-iteration node memo of bellman_vi / bellman_pi (bellman.c:1333-1353, 1773-1806) with the grid multi-index
 * packed into integers instead of printed into a string: same keys (node index + iteration counters), same hits
 * and misses, ~20x cheaper than snprintf + strcmp.  Used by the index-based batch entry points the solver loops
 * drive; the string-keyed HTable (bit-identical to hashgrid.c) stays behind the coordinate-based callbacks. */
struct FmEntry { uint64_t k[4]; double v; uint32_t epoch, pad; }; /* 48 bytes: key, value and liveness in one place */
struct FastMemo {
    size_t cap, used; /* cap is a power of two */
    struct FmEntry *e;
    uint32_t epoch;   /* entries of older epochs are empty: clearing is O(1) */
};

static uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
/* linear in the key words before the final mix, so a fiber can advance it per node with one multiply-add */
static const uint64_t FM_P[4] = {0x9e3779b97f4a7c15ULL, 0xc2b2ae3d27d4eb4fULL, 0x165667b19e3779f9ULL, 0xd6e8feb86659fd93ULL};
static uint64_t key_pre(const uint64_t k[4]) { return k[0] * FM_P[0] + k[1] * FM_P[1] + k[2] * FM_P[2] + k[3] * FM_P[3]; }

#define FM_CAP0 (1u << 16)
struct FastMemo *fastmemo_create(void)
{
    struct FastMemo *m = xmalloc(sizeof(*m));
    m->cap = FM_CAP0; m->used = 0; m->epoch = 1;
    m->e = calloc(m->cap, sizeof(*m->e));
    if (m->e == NULL) { fprintf(stderr, "c3sc: out of memory\n"); exit(1); }
    return m;
}
void fastmemo_free(struct FastMemo *m) { if (m) { free(m->e); free(m); } }
void fastmemo_clear(struct FastMemo *m)
{
    m->used = 0;
    if (++m->epoch == 0) { memset(m->e, 0, m->cap * sizeof(*m->e)); m->epoch = 1; } /* wrapped: really empty it */
}
size_t fastmemo_size(const struct FastMemo *m) { return m->used; }

static size_t fm_find_pre(const struct FastMemo *m, uint64_t pre, const uint64_t k[4])
{
    size_t i = (size_t)mix64(pre) & (m->cap - 1);
    while (m->e[i].epoch == m->epoch && memcmp(m->e[i].k, k, 4 * sizeof(uint64_t)) != 0) i = (i + 1) & (m->cap - 1);
    return i;
}
static void fm_put_pre(struct FastMemo *m, uint64_t pre, const uint64_t k[4], double val);
static void fm_grow(struct FastMemo *m)
{
    struct FastMemo old = *m;
    m->cap *= 2; m->used = 0; m->epoch = 1;
    m->e = calloc(m->cap, sizeof(*m->e));
    if (m->e == NULL) { fprintf(stderr, "c3sc: out of memory\n"); exit(1); }
    for (size_t i = 0; i < old.cap; i++)
        if (old.e[i].epoch == old.epoch) fm_put_pre(m, key_pre(old.e[i].k), old.e[i].k, old.e[i].v);
    free(old.e);
}
static void fm_put_pre(struct FastMemo *m, uint64_t pre, const uint64_t k[4], double val)
{ /* like htable_add_element, a repeated key is not checked by the callers; here the first value stays */
    if (2 * (m->used + 1) > m->cap) fm_grow(m);
    struct FmEntry *e = &m->e[fm_find_pre(m, pre, k)];
    if (e->epoch == m->epoch) return;
    memcpy(e->k, k, 4 * sizeof(uint64_t));
    e->v = val;
    e->epoch = m->epoch;
    m->used++;
}
int fastmemo_get(const struct FastMemo *m, const uint64_t k[4], double *val)
{
    const struct FmEntry *e = &m->e[fm_find_pre(m, key_pre(k), k)];
    if (e->epoch != m->epoch) return 0;
    *val = e->v;
    return 1;
}
void fastmemo_put(struct FastMemo *m, const uint64_t k[4], double val) { fm_put_pre(m, key_pre(k), k, val); }

/* grid multi-index (16 bits per dimension, d <= 12) + two counters */
void fastmemo_key(size_t d, const int32_t *idx, size_t kdim, size_t j, uint64_t c0, uint64_t c1, uint64_t key[4])
{
    key[0] = key[1] = key[2] = 0;
    for (size_t m = 0; m < d; m++) {
        const uint64_t v = (uint64_t)(m == kdim ? j : (size_t)idx[m]) & 0xffffu;
        key[m / 4] |= v << (16 * (m % 4));
    }
    key[3] = (c0 << 32) ^ c1;
}

/* the nodes of one fiber: the key and its pre-hash are built once, each node patches the 16 bits of the varying
 * dimension and adds j * (that field's weight) to the pre-hash */
void fastmemo_fiber_begin(struct FmFiber *ff, size_t d, const int32_t *idx, size_t kdim, uint64_t c0, uint64_t c1)
{
    fastmemo_key(d, idx, kdim, 0, c0, c1, ff->key);
    ff->word = (unsigned)(kdim / 4);
    ff->shift = (unsigned)(16 * (kdim % 4));
    ff->base = ff->key[ff->word];
    ff->pre0 = key_pre(ff->key);
    ff->step = FM_P[ff->word] << ff->shift;
}
void fastmemo_fiber_counter(struct FmFiber *ff, uint64_t c0, uint64_t c1)
{ /* change the two counters of the current fiber (e.g. the component of a stored vector) */
    const uint64_t nw = (c0 << 32) ^ c1;
    ff->pre0 += (nw - ff->key[3]) * FM_P[3];
    ff->key[3] = nw;
}
void fastmemo_fiber_prefetch(const struct FastMemo *m, const struct FmFiber *ff, size_t n)
{ /* the n home slots of the fiber's nodes are independent cache misses: start them all before the first probe */
    for (size_t j = 0; j < n; j++)
        __builtin_prefetch(&m->e[(size_t)mix64(ff->pre0 + (uint64_t)(j & 0xffffu) * ff->step) & (m->cap - 1)]);
}
int fastmemo_fiber_get(const struct FastMemo *m, struct FmFiber *ff, size_t j, double *val)
{
    ff->key[ff->word] = ff->base | ((uint64_t)(j & 0xffffu) << ff->shift);
    const struct FmEntry *e = &m->e[fm_find_pre(m, ff->pre0 + (uint64_t)(j & 0xffffu) * ff->step, ff->key)];
    if (e->epoch != m->epoch) return 0;
    *val = e->v;
    return 1;
}
void fastmemo_fiber_put(struct FastMemo *m, struct FmFiber *ff, size_t j, double val);
/* Lookup that also reports where the key would go (the first empty slot of its probe sequence), and an insert that uses
 * that slot if the table has not been rebuilt and the slot is still empty -- then it is still the first empty slot of the
 * sequence and the key cannot have been stored in between (it would sit there).  The batch entry points look every node up,
 * launch the misses and store them: the second probe sequence per stored node goes away. */
size_t fastmemo_cap(const struct FastMemo *m) { return m->cap; }
int fastmemo_fiber_get_slot(const struct FastMemo *m, struct FmFiber *ff, size_t j, double *val, size_t *slot)
{
    ff->key[ff->word] = ff->base | ((uint64_t)(j & 0xffffu) << ff->shift);
    const size_t i = fm_find_pre(m, ff->pre0 + (uint64_t)(j & 0xffffu) * ff->step, ff->key);
    *slot = i;
    if (m->e[i].epoch != m->epoch) return 0;
    *val = m->e[i].v;
    return 1;
}
void fastmemo_fiber_put_at(struct FastMemo *m, struct FmFiber *ff, size_t j, double val, size_t slot, size_t cap_then)
{
    if (m->cap != cap_then || 2 * (m->used + 1) > m->cap || m->e[slot].epoch == m->epoch) { fastmemo_fiber_put(m, ff, j, val); return; }
    ff->key[ff->word] = ff->base | ((uint64_t)(j & 0xffffu) << ff->shift);
    struct FmEntry *e = &m->e[slot];
    memcpy(e->k, ff->key, 4 * sizeof(uint64_t));
    e->v = val;
    e->epoch = m->epoch;
    m->used++;
}
void fastmemo_fiber_put(struct FastMemo *m, struct FmFiber *ff, size_t j, double val)
{
    ff->key[ff->word] = ff->base | ((uint64_t)(j & 0xffffu) << ff->shift);
    fm_put_pre(m, ff->pre0 + (uint64_t)(j & 0xffffu) * ff->step, ff->key, val);
}

/* ------------------------------------------------------------------------------ small helpers of util.h */
int c3sc_check_bounds(size_t dx, double *lbx, double *ubx, const double *x)
{ /* util.c:225-241: 0 inside (or no bounds), -(i+1) below the lower bound of dim i, +(i+1) above its upper bound */
    if (lbx == NULL || ubx == NULL) return 0;
    for (size_t i = 0; i < dx; i++) {
        if (x[i] < lbx[i]) return -(int)(i + 1);
        if (x[i] > ubx[i]) return (int)(i + 1);
    }
    return 0;
}

static int cmp_double(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

double *c3sc_combine_and_sort(size_t Nx, double *x, size_t Ny, double *y, size_t *Ntot)
{ /* util.c:254-274: sorted union, values within 1e-15 of the last kept one are dropped; the caller frees */
    const size_t n = Nx + Ny;
    double *all = xmalloc((n ? n : 1) * sizeof(double));
    if (Nx) memcpy(all, x, Nx * sizeof(double));
    if (Ny) memcpy(all + Nx, y, Ny * sizeof(double));
    qsort(all, n, sizeof(double), cmp_double);
    size_t kept = 0;
    for (size_t i = 0; i < n; i++)
        if (kept == 0 || fabs(all[i] - all[kept - 1]) > 1e-15) all[kept++] = all[i];
    for (size_t i = kept; i < n; i++) all[i] = 0.0;
    *Ntot = kept;
    return all;
}

struct ProbInd { double p; size_t ind; };
static int cmp_probind(const void *a, const void *b)
{
    const double x = ((const struct ProbInd *)a)->p, y = ((const struct ProbInd *)b)->p;
    return (x > y) - (x < y);
}

size_t c3sc_sample_discrete_rv(size_t n, double *probs, double sample)
{ /* util.c:299-331: sort ascending, overwrite probs with the running sums, first bin whose sum reaches the sample */
    if (n > 1000) { fprintf(stderr, "Not enough memory allocate in discrete_sample\n"); exit(1); }
    struct ProbInd sc[1000];
    for (size_t i = 0; i < n; i++) { sc[i].p = probs[i]; sc[i].ind = i; }
    qsort(sc, n, sizeof(sc[0]), cmp_probind);
    double run = 0.0;
    for (size_t i = 0; i < n; i++) { run = (i == 0) ? sc[0].p : sc[i].p + run; probs[i] = run; }
    for (size_t i = 0; i < n; i++) if (sample <= probs[i]) return sc[i].ind;
    fprintf(stderr, "Warning: problem with gen. sample\nUniform sample is %G\n", sample);
    return 0;
}

/* ------------------------------------------------------------------------------ HashGrid: grid value -> node index
 * (util.c:352-657).  The reference keys on C3's text serialisation of the double, i.e. on the exact value; here the
 * key is the bit pattern (-0.0 folded into +0.0), chained per bucket like the original. */
struct HGNode { uint64_t bits; size_t ind; struct HGNode *next; };
struct HashGrid { size_t size; struct HGNode **table; };

static uint64_t hg_bits(double v)
{
    if (v == 0.0) v = 0.0;
    uint64_t b;
    memcpy(&b, &v, sizeof(b));
    return b;
}

struct HashGrid *hash_grid_create(size_t size)
{
    if (size < 1) return NULL;
    struct HashGrid *h = xmalloc(sizeof(*h));
    h->size = size;
    h->table = xcalloc(size, sizeof(*h->table));
    return h;
}
void hash_grid_free(struct HashGrid *h)
{
    if (h == NULL) return;
    for (size_t i = 0; i < h->size; i++)
        for (struct HGNode *n = h->table[i]; n != NULL;) { struct HGNode *nx = n->next; free(n); n = nx; }
    free(h->table);
    free(h);
}
static struct HGNode *hg_find(const struct HashGrid *h, uint64_t bits)
{
    for (struct HGNode *n = h->table[mix64(bits) % h->size]; n != NULL; n = n->next)
        if (n->bits == bits) return n;
    return NULL;
}
int hash_grid_add_element(struct HashGrid *h, size_t ind, double val)
{ /* 0 added, 2 the value is already there (util.c:527-563) */
    const uint64_t bits = hg_bits(val);
    if (hg_find(h, bits) != NULL) return 2;
    struct HGNode *n = xmalloc(sizeof(*n));
    n->bits = bits; n->ind = ind;
    struct HGNode **head = &h->table[mix64(bits) % h->size];
    n->next = *head;
    *head = n;
    return 0;
}
size_t hash_grid_get_ind(struct HashGrid *h, double val, int *exists)
{ /* index of the value, *exists = 1; a missing value gives 0 with *exists = 0 (util.c:589-615) */
    const struct HGNode *n = hg_find(h, hg_bits(val));
    *exists = n != NULL;
    return n ? n->ind : 0;
}
struct HashGrid *hash_grid_create_grid(size_t size, const struct c3Vector *grid)
{
    struct HashGrid *h = hash_grid_create(size);
    for (size_t i = 0; h != NULL && i < grid->size; i++) hash_grid_add_element(h, i, grid->elem[i]);
    return h;
}
struct HashGrid **hash_grid_create_ndgrid(size_t size, size_t d, struct c3Vector **grid)
{
    struct HashGrid **hs = xcalloc(d, sizeof(*hs));
    for (size_t m = 0; m < d; m++) hs[m] = hash_grid_create_grid(size, grid[m]);
    return hs;
}
void hash_grid_free_ndgrid(size_t d, struct HashGrid **hs)
{
    if (hs == NULL) return;
    for (size_t m = 0; m < d; m++) hash_grid_free(hs[m]);
    free(hs);
}
int hash_grid_ndgrid_get_ind(struct HashGrid **hs, size_t dim, const double *x, size_t *out)
{ /* 0 every coordinate found, 1 otherwise (stops at the first miss, util.c:623-641) */
    for (size_t m = 0; m < dim; m++) {
        int ok = 0;
        out[m] = hash_grid_get_ind(hs[m], x[m], &ok);
        if (!ok) return 1;
    }
    return 0;
}
void hash_grid_print(struct HashGrid *h, FILE *fp)
{ /* one line per occupied bucket (util.c:566-580) */
    for (size_t i = 0; i < h->size; i++) {
        if (h->table[i] == NULL) continue;
        for (const struct HGNode *n = h->table[i]; n != NULL; n = n->next) {
            double v;
            memcpy(&v, &n->bits, sizeof(v));
            fprintf(fp, "ind=%zu,val = %3.15G ", i, v);
        }
        fprintf(fp, "\n");
    }
}

/* ------------------------------------------------------------------------------ Workspace */
struct Workspace {
    size_t dx, du, dw, N;
    size_t off[11]; /* slab layout util.c:738-748 */
    double *slab;   /* N nodes x off[10] doubles */
    double *costs;  /* N x (2dx+1) */
    int *absorbed;
    size_t *absorbed_no, *absorbed_yes; /* util.c:702-703: index lists nobody reads any more */
    size_t *ind_to_serialize;
    struct HTable *vi_htable;
    size_t vi_iter;
    char **keys, **keys2;
    struct c3sc_hip_ctx *hip, *hip_policy;
    /* policy iteration (util.c:700-715, 766-779) */
    struct HTable *pi_prob_htable, *pi_htable;
    size_t pi_iter, pi_subiter;
    struct FastMemo *vi_fast, *pi_prob_fast;
};

#define NBUCKET 1000000 /* util.c:760 */

struct Workspace *workspace_alloc(size_t dx, size_t du, size_t dw, size_t N)
{
    struct Workspace *w = xmalloc(sizeof(*w));
    w->dx = dx; w->du = du; w->dw = dw; w->N = N;
    const size_t sz[11] = {dx, dx * du, dx * dw, dx * dw * du, 1, du, 2 * dx + 1, du * (2 * dx + 1), du, du, du};
    size_t acc = 0;
    for (int i = 0; i < 11; i++) { acc += sz[i]; w->off[i] = acc; }
    w->slab = xmalloc(N * w->off[10] * sizeof(double));
    w->costs = xmalloc(N * (2 * dx + 1) * sizeof(double));
    w->absorbed = xmalloc(N * sizeof(int));
    w->absorbed_no = xcalloc(N, sizeof(size_t));
    w->absorbed_yes = xcalloc(N, sizeof(size_t));
    w->ind_to_serialize = xmalloc((dx + 3) * sizeof(size_t));
    w->vi_htable = htable_create(NBUCKET);
    w->keys = xmalloc(N * sizeof(char *));
    w->keys2 = xmalloc(N * sizeof(char *));
    for (size_t i = 0; i < N; i++) { w->keys[i] = xmalloc(256); w->keys2[i] = xmalloc(256); }
    w->pi_prob_htable = htable_create(NBUCKET);
    w->pi_htable = htable_create(NBUCKET);
    w->vi_fast = fastmemo_create();
    w->pi_prob_fast = fastmemo_create();
    return w;
}

void workspace_free(struct Workspace *w)
{
    if (w == NULL) return;
    for (size_t i = 0; i < w->N; i++) { free(w->keys[i]); free(w->keys2[i]); }
    free(w->keys2);
    htable_destroy(w->pi_prob_htable); htable_destroy(w->pi_htable);
    fastmemo_free(w->vi_fast); fastmemo_free(w->pi_prob_fast);
    if (w->hip_policy) { c3sc_forget_ctx(w->hip_policy); c3sc_hip_ctx_destroy(w->hip_policy); }
    free(w->keys); free(w->slab); free(w->costs); free(w->absorbed); free(w->absorbed_no); free(w->absorbed_yes); free(w->ind_to_serialize);
    htable_destroy(w->vi_htable);
    if (w->hip) { c3sc_forget_ctx(w->hip); c3sc_hip_ctx_destroy(w->hip); }
    free(w);
}

void workspace_reset_vi_htable(struct Workspace *w)
{
    htable_destroy(w->vi_htable);
    w->vi_htable = htable_create(NBUCKET);
    fastmemo_clear(w->vi_fast);
}
struct FastMemo *workspace_get_vi_fastmemo(const struct Workspace *w) { return w->vi_fast; }
struct FastMemo *workspace_get_pi_prob_fastmemo(const struct Workspace *w) { return w->pi_prob_fast; }
void workspace_increment_vi_iter(struct Workspace *w) { w->vi_iter++; }
size_t workspace_get_vi_iter(const struct Workspace *w) { return w->vi_iter; }
struct HTable *workspace_get_vi_htable(const struct Workspace *w) { return w->vi_htable; }

static double *slot(struct Workspace *w, size_t node, int which)
{
    return w->slab + node * w->off[10] + (which == 0 ? 0 : w->off[which - 1]);
}
double *workspace_get_drift(struct Workspace *w, size_t n) { return slot(w, n, 0); }
double *workspace_get_grad_drift(struct Workspace *w, size_t n) { return slot(w, n, 1); }
double *workspace_get_diff(struct Workspace *w, size_t n) { return slot(w, n, 2); }
double *workspace_get_grad_diff(struct Workspace *w, size_t n) { return slot(w, n, 3); }
double *workspace_get_dt(struct Workspace *w, size_t n) { return slot(w, n, 4); }
double *workspace_get_grad_dt(struct Workspace *w, size_t n) { return slot(w, n, 5); }
double *workspace_get_prob(struct Workspace *w, size_t n) { return slot(w, n, 6); }
double *workspace_get_grad_prob(struct Workspace *w, size_t n) { return slot(w, n, 7); }
double *workspace_get_grad_stage(struct Workspace *w, size_t n) { return slot(w, n, 8); }
double *workspace_get_control_size_extra(struct Workspace *w, size_t n) { return slot(w, n, 9); }
double *workspace_get_u(struct Workspace *w, size_t n) { return slot(w, n, 10); }
double *workspace_get_costs(struct Workspace *w, size_t n) { return w->costs + n * (2 * w->dx + 1); }
int *workspace_get_absorbed(struct Workspace *w, size_t n) { return w->absorbed + n; }
size_t *workspace_get_ind_to_serialize(struct Workspace *w) { return w->ind_to_serialize; }
size_t *workspace_get_absorbed_no(struct Workspace *w) { return w->absorbed_no; }
size_t *workspace_get_absorbed_yes(struct Workspace *w) { return w->absorbed_yes; }
char **workspace_get_saved_keys(struct Workspace *w) { return w->keys; }
char **workspace_get_saved_keys2(struct Workspace *w) { return w->keys2; }
void workspace_reset_pi_prob_htable(struct Workspace *w)
{
    htable_destroy(w->pi_prob_htable);
    w->pi_prob_htable = htable_create(NBUCKET);
    fastmemo_clear(w->pi_prob_fast);
}
void workspace_reset_pi_htable(struct Workspace *w) { htable_destroy(w->pi_htable); w->pi_htable = htable_create(NBUCKET); }
struct HTable *workspace_get_pi_prob_htable(const struct Workspace *w) { return w->pi_prob_htable; }
struct HTable *workspace_get_pi_htable(const struct Workspace *w) { return w->pi_htable; }
void workspace_increment_pi_iter(struct Workspace *w) { w->pi_iter++; }
void workspace_increment_pi_subiter(struct Workspace *w) { w->pi_subiter++; }
size_t workspace_get_pi_iter(const struct Workspace *w) { return w->pi_iter; }
size_t workspace_get_pi_subiter(const struct Workspace *w) { return w->pi_subiter; }

static struct c3sc_hip_ctx *make_ctx(void)
{
    struct c3sc_hip_ctx *ctx = NULL;
    const char *dev = getenv("C3SC_HIP_DEVICE");
    int rc = c3sc_hip_ctx_create(dev ? atoi(dev) : 0, &ctx);
    if (rc != C3SC_OK) { /* no CPU fallback: the reference style is to print and exit(1) */
        fprintf(stderr, "c3sc: cannot create the MI355X context (code %d); the Bellman backup has no CPU fallback\n", rc);
        exit(1);
    }
    return ctx;
}

/* the contexts if they exist already (no creation): for end-of-sweep status checks */
struct c3sc_hip_ctx *workspace_peek_hip_ctx(struct Workspace *w) { return w->hip; }
struct c3sc_hip_ctx *workspace_peek_hip_ctx_policy(struct Workspace *w) { return w->hip_policy; }

struct c3sc_hip_ctx *workspace_get_hip_ctx(struct Workspace *w)
{
    if (w->hip == NULL) w->hip = make_ctx();
    return w->hip;
}

/* second engine holding the POLICY's value function during policy iteration (bellman_pi reads two value
 * functions per fiber, bellman.c:1741 and :1767) */
struct c3sc_hip_ctx *workspace_get_hip_ctx_policy(struct Workspace *w)
{
    if (w->hip_policy == NULL) w->hip_policy = make_ctx();
    return w->hip_policy;
}
