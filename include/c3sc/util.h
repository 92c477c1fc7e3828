/* util.h -- ApproxArgs, HashGrid, Workspace (mirrors src/util.h:49-125) and the
 * minimal brute-force optimiser object that stands where C3's c3Opt stands in the reference. */
#ifndef C3SC_UTIL_H
#define C3SC_UTIL_H
#include <stddef.h>

#include "hashgrid.h"

/* C3's enum function_class is reduced to the two classes valuefunc.c:655-670 accepts */
enum function_class { CONSTELM = 4, LINELM = 5 };

struct ApproxArgs;
struct ApproxArgs *approx_args_init(void); /* defaults util.c:124-130 */
void approx_args_free(struct ApproxArgs *);
void approx_args_set_function_class(struct ApproxArgs *, enum function_class);
enum function_class approx_args_get_function_class(const struct ApproxArgs *);
void approx_args_set_cross_tol(struct ApproxArgs *, double);
double approx_args_get_cross_tol(const struct ApproxArgs *);
void approx_args_set_round_tol(struct ApproxArgs *, double);
double approx_args_get_round_tol(const struct ApproxArgs *);
void approx_args_set_kickrank(struct ApproxArgs *, size_t);
size_t approx_args_get_kickrank(const struct ApproxArgs *);
void approx_args_set_maxrank(struct ApproxArgs *, size_t);
size_t approx_args_get_maxrank(const struct ApproxArgs *);
void approx_args_set_startrank(struct ApproxArgs *, size_t);
size_t approx_args_get_startrank(const struct ApproxArgs *);
void approx_args_set_adapt(struct ApproxArgs *, int);
int approx_args_get_adapt(const struct ApproxArgs *);
/* new (no reference counterpart): ranks of the cross approximation before its result is rounded to maxrank; 0 = maxrank */
void approx_args_set_crossrank(struct ApproxArgs *, size_t);
size_t approx_args_get_crossrank(const struct ApproxArgs *);
/* new: cap on the cross iterations of one interpolation (the reference hard-codes 5: valuefunc.c:632) */
void approx_args_set_cross_maxiter(struct ApproxArgs *, size_t);
size_t approx_args_get_cross_maxiter(const struct ApproxArgs *);
size_t uniform_stride(size_t N, size_t M); /* util.c:995-1006 */

#include <stdio.h>
int c3sc_check_bounds(size_t dx, double *lbx, double *ubx, const double *x);                   /* util.c:225-241 */
size_t c3sc_sample_discrete_rv(size_t n, double *probs, double sample);                       /* util.c:299-331; probs is overwritten */
double *c3sc_combine_and_sort(size_t Nx, double *x, size_t Ny, double *y, size_t *Ntot);      /* util.c:254-274; caller frees */

/* C3's array.h vector (size, elem) as far as the HashGrid constructors read it */
struct c3Vector { size_t size; double *elem; };
/* grid value -> node index by exact value (util.c:352-657) */
struct HashGrid;
struct HashGrid *hash_grid_create(size_t size);
struct HashGrid *hash_grid_create_grid(size_t size, const struct c3Vector *grid);
struct HashGrid **hash_grid_create_ndgrid(size_t size, size_t d, struct c3Vector **grid);
void hash_grid_free_ndgrid(size_t d, struct HashGrid **);
int hash_grid_add_element(struct HashGrid *, size_t ind, double val);                        /* 0 added, 2 already present */
void hash_grid_print(struct HashGrid *, FILE *);
size_t hash_grid_get_ind(struct HashGrid *, double val, int *exists);
int hash_grid_ndgrid_get_ind(struct HashGrid **, size_t dim, const double *x, size_t *out);  /* 0 found, 1 not */
void hash_grid_free(struct HashGrid *);

/* ---- brute-force c3Opt subset (C3 lib_optimization.h names, own implementation) ---- */
enum c3opt_alg { BFGS = 0, LBFGS = 1, BATCHGRAD = 2, BRUTEFORCE = 3 };
struct c3Opt;
struct c3Opt *c3opt_alloc(enum c3opt_alg alg, size_t d); /* BRUTEFORCE: list scan; others: box grid + golden-section polish */
struct c3Opt *c3opt_copy(struct c3Opt *);
void c3opt_free(struct c3Opt *);
void c3opt_set_brute_force_vals(struct c3Opt *, size_t n, double *vals /* n x d */);
int c3opt_is_bruteforce(const struct c3Opt *);
void c3opt_add_objective(struct c3Opt *, double (*f)(size_t, const double *, double *, void *), void *arg);
int c3opt_minimize(struct c3Opt *, double *x, double *val); /* scan in order, strict '<' */
size_t c3opt_get_nbrute(const struct c3Opt *);
const double *c3opt_get_brute_vals(const struct c3Opt *);
size_t c3opt_get_d(const struct c3Opt *);
void c3opt_add_lb(struct c3Opt *, double *lb);
void c3opt_add_ub(struct c3Opt *, double *ub);
double *c3opt_get_lb(struct c3Opt *);
double *c3opt_get_ub(struct c3Opt *);
void c3opt_set_relftol(struct c3Opt *, double);  /* the line-search knobs of C3's optimisers are accepted and ignored */
void c3opt_set_absxtol(struct c3Opt *, double);
void c3opt_set_gtol(struct c3Opt *, double);
void c3opt_set_maxiter(struct c3Opt *, size_t);
void c3opt_ls_set_maxiter(struct c3Opt *, size_t);
void c3opt_ls_set_alpha(struct c3Opt *, double);
void c3opt_ls_set_beta(struct c3Opt *, double);
void c3opt_set_verbose(struct c3Opt *, int);
void c3opt_set_storage_options(struct c3Opt *, int, int, int);
void c3opt_set_box_search(struct c3Opt *, size_t grid, size_t polish); /* new */
size_t c3opt_get_box_grid(const struct c3Opt *);
size_t c3opt_get_box_polish(const struct c3Opt *);

/* ---- Workspace (util.c:689-964): per-node scratch, key buffers, memo tables, + the device context ---- */
struct Workspace;
struct Workspace *workspace_alloc(size_t dx, size_t du, size_t dw, size_t N);
void workspace_free(struct Workspace *);
void workspace_reset_vi_htable(struct Workspace *);
void workspace_increment_vi_iter(struct Workspace *);
size_t workspace_get_vi_iter(const struct Workspace *);
struct HTable *workspace_get_vi_htable(const struct Workspace *);
double *workspace_get_drift(struct Workspace *, size_t node);
double *workspace_get_grad_drift(struct Workspace *, size_t node);
double *workspace_get_diff(struct Workspace *, size_t node);
double *workspace_get_grad_diff(struct Workspace *, size_t node);
double *workspace_get_dt(struct Workspace *, size_t node);
double *workspace_get_grad_dt(struct Workspace *, size_t node);
double *workspace_get_prob(struct Workspace *, size_t node);
double *workspace_get_grad_prob(struct Workspace *, size_t node);
double *workspace_get_grad_stage(struct Workspace *, size_t node);
double *workspace_get_control_size_extra(struct Workspace *, size_t node);
double *workspace_get_u(struct Workspace *, size_t node);
double *workspace_get_costs(struct Workspace *, size_t node);
int *workspace_get_absorbed(struct Workspace *, size_t node);
size_t *workspace_get_ind_to_serialize(struct Workspace *);
size_t *workspace_get_absorbed_no(struct Workspace *);  /* N-entry scratch lists (util.c:946-954) */
size_t *workspace_get_absorbed_yes(struct Workspace *);
char **workspace_get_saved_keys(struct Workspace *);
char **workspace_get_saved_keys2(struct Workspace *);
/* policy iteration state (util.c:700-715, 766-779, 930-964) */
void workspace_reset_pi_prob_htable(struct Workspace *);
void workspace_reset_pi_htable(struct Workspace *);
struct HTable *workspace_get_pi_prob_htable(const struct Workspace *);
struct HTable *workspace_get_pi_htable(const struct Workspace *);
void workspace_increment_pi_iter(struct Workspace *);
void workspace_increment_pi_subiter(struct Workspace *);
size_t workspace_get_pi_iter(const struct Workspace *);
size_t workspace_get_pi_subiter(const struct Workspace *);
/* new: integer-keyed twin of the node memo (same keys, no strings) behind the index-based batch entry points */
#include <stdint.h>
struct FastMemo;
struct FastMemo *fastmemo_create(void);
void fastmemo_free(struct FastMemo *);
void fastmemo_clear(struct FastMemo *);
size_t fastmemo_size(const struct FastMemo *);
int fastmemo_get(const struct FastMemo *, const uint64_t key[4], double *val);
void fastmemo_put(struct FastMemo *, const uint64_t key[4], double val);
void fastmemo_key(size_t d, const int32_t *idx, size_t kdim, size_t j, uint64_t c0, uint64_t c1, uint64_t key[4]);
/* the same keys for the nodes j of one fiber along kdim, built incrementally */
struct FmFiber { uint64_t key[4], base, pre0, step; unsigned word, shift; };
void fastmemo_fiber_begin(struct FmFiber *ff, size_t d, const int32_t *idx, size_t kdim, uint64_t c0, uint64_t c1);
void fastmemo_fiber_counter(struct FmFiber *ff, uint64_t c0, uint64_t c1);
void fastmemo_fiber_prefetch(const struct FastMemo *, const struct FmFiber *ff, size_t n); /* warm the n home slots */
int fastmemo_fiber_get(const struct FastMemo *, struct FmFiber *ff, size_t j, double *val);
void fastmemo_fiber_put(struct FastMemo *, struct FmFiber *ff, size_t j, double val);
size_t fastmemo_cap(const struct FastMemo *);
int fastmemo_fiber_get_slot(const struct FastMemo *, struct FmFiber *ff, size_t j, double *val, size_t *slot); /* + where the key would go */
void fastmemo_fiber_put_at(struct FastMemo *, struct FmFiber *ff, size_t j, double val, size_t slot, size_t cap_then); /* insert there if still valid */
struct FastMemo *workspace_get_vi_fastmemo(const struct Workspace *);
struct FastMemo *workspace_get_pi_prob_fastmemo(const struct Workspace *);
/* new: the MI355X engine this workspace drives (created on first use; aborts if no GPU) */
struct c3sc_hip_ctx;
struct c3sc_hip_ctx *workspace_get_hip_ctx(struct Workspace *);
struct c3sc_hip_ctx *workspace_get_hip_ctx_policy(struct Workspace *); /* holds the policy's value function */
struct c3sc_hip_ctx *workspace_peek_hip_ctx(struct Workspace *);        /* NULL until the first device call */
struct c3sc_hip_ctx *workspace_peek_hip_ctx_policy(struct Workspace *);
#endif
