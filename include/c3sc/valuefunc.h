/* valuefunc.h -- nodal function-train value function (mirrors src/valuefunc.h:47-76 for the hot path).
 * The reference keeps a C3 FunctionTrain inside struct ValueF; C3 is out of scope here, so the value
 * function is held directly as its nodal core tables (exactly what valuef_precompute_cores,
 * valuefunc.c:165-189, produces), its grid and the cross index sets of the interpolation that made it.
 * valuef_interp and the norms are an own restatement of what the reference delegates to C3 (c3sc_cross.c). */
#ifndef C3SC_VALUE_H
#define C3SC_VALUE_H
#include <stddef.h>

struct ValueF;
/* new constructor: cores[m][j*r_m*r_{m+1} + a + b*r_m], ranks[d+1]; copies its inputs */
struct ValueF *valuef_create_nodal(size_t d, const size_t *N, const size_t *ranks, double **cores);
void valuef_destroy(struct ValueF *);
struct ValueF *valuef_copy(struct ValueF *);
size_t *valuef_get_ranks(struct ValueF *);
struct CrossIndex;                                     /* opaque: the left index sets of the last interpolation */
struct CrossIndex **valuef_get_isl(const struct ValueF *);   /* valuefunc.c:218-221 */
size_t valuef_get_dim(const struct ValueF *);
const size_t *valuef_get_N(const struct ValueF *);
double **valuef_get_cores(struct ValueF *);
/* value at a grid multi-index (what valuef_eval returns at grid nodes, valuefunc.c:345-350) */
double valuef_eval_ind(struct ValueF *, const size_t *ind);
/* valuefunc.c:369-585 -- runs on the GPU bound with valuef_bind_device */
int valuef_eval_fiber_ind_nn(struct ValueF *, const size_t *fixed_ind, size_t dim_vary, const size_t *neighbors,
                             const size_t *neighbors_vary, double *out);
/* the nodes the cores are sampled on (copied); valuef_interp attaches them itself */
void valuef_attach_grid(struct ValueF *, double **grid);
/* valuefunc.c:603-767: cross approximation of a fiber callback f(N, x[N*d], out[N], args) on the tensor grid.
 * vref (may be NULL) seeds ranks (+1) and index sets when aargs->adapt == 1. */
struct ApproxArgs;
struct ValueF *valuef_interp(size_t d, int (*f)(size_t, const double *, double *, void *), void *args, const size_t *N,
                             double **grid, struct ValueF *vref, struct ApproxArgs *aargs, int verbose);
/* new: the same with a batched callback fb(F, N, x[F*N*d], out[F*N], args): all r_k r_{k+1} fibers of a core step
 * arrive in one call (bellman_vi_batch / bellman_pi_batch run them in one kernel launch) */
struct ValueF *valuef_interp_batch(size_t d, int (*fb)(size_t, size_t, const double *, double *, void *), void *args,
                                   const size_t *N, double **grid, struct ValueF *vref, struct ApproxArgs *aargs,
                                   int verbose);
/* new: the callback receives grid indices instead of coordinates: fi(F, dim_vary, idx[F*d] (entry dim_vary ignored),
 * out[F*N], args) -- bellman_vi_batch_idx / bellman_pi_batch_idx */
#include <stdint.h>
struct ValueF *valuef_interp_idx(size_t d, int (*fi)(size_t, size_t, const int32_t *, double *, void *), void *args,
                                 const size_t *N, double **grid, struct ValueF *vref, struct ApproxArgs *aargs,
                                 int verbose);
/* new: the same, with the fibers of every core step sharded over `world` ranks (one process per GPU): rank r runs fi on
 * the contiguous block [lo, hi) = ceil(F/world) fibers and `exchange(out, F, N_k, lo, hi, xarg)` fills the other ranks'
 * rows of out[F*N_k] (an all-gather: RCCL on the GPU box, gloo in the CPU tests; c3sc_amd/distributed.py builds it).
 * Every rank then holds the same fiber values and takes the same pivot decisions, so the ranks' results are bit-identical to each
 * other; they equal the unsharded call's bit for bit when fi is a function of the node (a C3Control's default, see
 * c3control_set_consistent_ends) or, with the literal end-point rule, when the caller keeps every rank's node memo complete
 * (c3control_step_vi does: the rows other ranks computed enter its memo; c3control_step_pi's per-node policy memo holds this
 * rank's rows only, so with the literal rule a sharded policy evaluation may break a near-tie differently from the unsharded
 * run).  A rank whose fi fails still enters the exchange (its rows marked NaN) and all ranks return the error together.
 * SURVEY.md 8e; the reference hook is bellman.c:2201. */
typedef int (*c3sc_exchange_fn)(double *out, size_t F, size_t N, size_t lo, size_t hi, void *xarg);
/* diagnostics of this library's cross driver (no reference counterpart): 0 = sweeps whose first iteration was tried speculatively
 * on the device (c3sc_hip_cross_speculate: a sweep whose predecessor ended with the index sets it started from), 1 = those that
 * were confirmed, i.e. whose whole iteration took d + 1 kernel launches; process-wide counts */
size_t valuef_interp_counter(int which);
struct ValueF *valuef_interp_idx_sharded(size_t d, int (*fi)(size_t, size_t, const int32_t *, double *, void *), void *args,
                                         const size_t *N, double **grid, struct ValueF *vref, struct ApproxArgs *aargs,
                                         int verbose, size_t world, size_t rank, c3sc_exchange_fn exchange, void *xarg);
double valuef_norm(struct ValueF *);                       /* valuefunc.c:315-322: sqrt(int V^2), linear elements */
double valuef_norm2diff(struct ValueF *, struct ValueF *); /* valuefunc.c:324-335 */
double valuef_eval(struct ValueF *, const double *x);      /* valuefunc.c:337-343: off-grid multilinear interpolant */
/* valuefunc.c:226-295 in an own file format (the .c3 bytes live in C3): save returns 0 on success, load returns NULL
 * if the file cannot be opened; ngrid/xgrid (may be NULL) give the grid to resample onto */
int valuef_save(struct ValueF *, char *filename);
struct ValueF *valuef_load(char *filename, size_t *ngrid, double **xgrid);
int valuef_savetxt(struct ValueF *, char *filename);
struct ValueF *valuef_loadtxt(char *filename, size_t *ngrid, double **xgrid);
struct c3sc_hip_ctx;
void valuef_bind_device(struct ValueF *, struct c3sc_hip_ctx *); /* uploads the cores (lazy, once per ctx) */
#endif
