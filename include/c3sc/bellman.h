/* bellman.h -- Bellman backup API (mirrors src/bellman.h:48-194 for the value-iteration path).
 * bellman_vi keeps the callback ABI C3's cross approximation uses (valuefunc.c:615-616) and runs
 * the fiber on the MI355X through include/c3sc_hip.h. */
#ifndef C3SC_BELLMAN_H
#define C3SC_BELLMAN_H
#include <stddef.h>

#include "boundary.h"
#include "dynamics.h"
#include "nodeutil.h"
#include "util.h"
#include "valuefunc.h"

double bellmanrhs(size_t dx, size_t du, double stage_cost, const double *stage_grad, double discount,
                  const double *prob, const double *prob_grad, double dt, const double *dtgrad, const double *cost,
                  double *grad); /* bellman.c:88-112 */

struct MCAparam;
struct MCAparam *mca_param_create(size_t dx, size_t du);
void mca_add_grid_refs(struct MCAparam *, size_t *ngrid, double **xgrid, double hmin, double *hvec); /* bellman.c:171-188 */
void mca_param_destroy(struct MCAparam *);

struct DPparam;
struct DPparam *dp_param_create(size_t dx, size_t du, size_t dw, double discount);
void dp_param_destroy(struct DPparam *);
void dp_param_add_drift(struct DPparam *, c3sc_dyn_fn, void *);
void dp_param_add_diff(struct DPparam *, c3sc_dyn_fn, void *);
void dp_param_add_boundary(struct DPparam *, struct Boundary *);
void dp_param_add_stagecost(struct DPparam *, int (*)(double, const double *, const double *, double *, double *));
void dp_param_add_boundcost(struct DPparam *, int (*)(double, const double *, double *));
void dp_param_add_obscost(struct DPparam *, int (*)(const double *, double *));
/* new (opt-in): the device functor (include/c3sc_hip.h C3SC_MODEL_*) that restates the callbacks above for the
 * kernels.  bellman_vi cross-checks it against the host callbacks on the first fiber it runs. */
void dp_param_set_device_model(struct DPparam *, int model, const double *params, size_t nparams);

struct ControlParams;
struct ControlParams *control_params_create(size_t dx, size_t dw, struct DPparam *, struct MCAparam *, struct Workspace *,
                                            struct c3Opt *);
void control_params_add_time_and_states(struct ControlParams *, double time, size_t N, const double *x);
int control_params_get_last_res(const struct ControlParams *);
void control_params_destroy(struct ControlParams *);

double bellman_control(size_t du, const double *u, double *grad_u, void *args); /* bellman.c:367-480 */
int bellman_optimal(size_t du, double *u, double *val, void *arg);             /* bellman.c:504-543 (BRUTEFORCE) */

struct VIparam;
struct VIparam *vi_param_create(double convergence);
void vi_param_destroy(struct VIparam *);
void vi_param_add_cp(struct VIparam *, struct ControlParams *);
void vi_param_add_value(struct VIparam *, struct ValueF *);
size_t vi_param_get_nnode_evals(const struct VIparam *);
int bellman_vi(size_t N, const double *x, double *out, void *arg); /* bellman.c:1295-1423 */

/* new: many fibers in one launch (what an own cross driver would call per core step); x is F blocks of
 * N x dx; memo semantics identical to F successive bellman_vi calls */
int bellman_vi_batch(size_t F, size_t N, const double *x, double *out, void *arg);
/* new: fibers by grid indices idx[F][dx] along dim k (entry k ignored); integer-keyed twin of the memo */
#include <stdint.h>
int bellman_vi_batch_idx(size_t F, size_t k, const int32_t *idx, double *out, void *arg);

/* policy evaluation (bellman.c:1430-1491, 1702-1886): vf_policy fixes the control at every node, the Bellman
 * right-hand side is evaluated on vf_iteration */
struct PIparam;
struct PIparam *pi_param_create(double convergence, struct ValueF *policy);
void pi_param_destroy(struct PIparam *);
void pi_param_add_cp(struct PIparam *, struct ControlParams *);
void pi_param_add_value(struct PIparam *, struct ValueF *);
size_t pi_param_get_npol_evals(const struct PIparam *);       /* new: read-only views of the reference's counters */
size_t pi_param_get_niter_node_evals(const struct PIparam *);
int bellman_pi(size_t N, const double *x, double *out, void *arg); /* bellman.c:1702-1886 */
int bellman_pi_batch(size_t F, size_t N, const double *x, double *out, void *arg); /* new: many fibers per launch */
int bellman_pi_batch_idx(size_t F, size_t k, const int32_t *idx, double *out, void *arg); /* new: by grid indices */

struct C3Control;
struct C3Control *c3control_create(size_t dx, size_t du, size_t dw, double *lb, double *ub, size_t *ngrid,
                                   double discount); /* bellman.c:1962-1999 */
void c3control_destroy(struct C3Control *);
size_t *c3control_get_ngrid(struct C3Control *);
double **c3control_get_xgrid(struct C3Control *);
struct Boundary *c3control_get_boundary(struct C3Control *); /* new: read access for callers of the nodeutil functions */
void c3control_set_external_boundary(struct C3Control *, size_t dim, char *type);
void c3control_add_obstacle(struct C3Control *, double *center, double *widths);
void c3control_add_drift(struct C3Control *, c3sc_dyn_fn, void *);
void c3control_add_diff(struct C3Control *, c3sc_dyn_fn, void *);
void c3control_add_stagecost(struct C3Control *, int (*)(double, const double *, const double *, double *, double *));
void c3control_add_boundcost(struct C3Control *, int (*)(double, const double *, double *));
void c3control_add_obscost(struct C3Control *, int (*)(const double *, double *));
void c3control_set_device_model(struct C3Control *, int model, const double *params, size_t nparams); /* new */
/* new: ON by default for a C3Control.  The end points of a reflecting / periodic fiber keep the absorbed flag the fixed
 * dimensions and obstacles give them, so a node's value does not depend on the direction of the fiber that computes it
 * (process_fibers_neighbor resets them, nodeutil.c:570-612: the reference's memo then keeps whichever direction came first).
 * 0, or C3SC_LITERAL_ENDS=1 in the environment, restores the literal rule. */
void c3control_set_consistent_ends(struct C3Control *, int on);
/* new: multi-GPU (one process per GPU, every rank runs the same solver): the fibers of every core step of step_vi /
 * step_pi are split over `world` ranks and all-gathered by `exchange` (valuefunc.h: valuef_interp_idx_sharded) */
void c3control_set_fiber_sharding(struct C3Control *, size_t world, size_t rank, c3sc_exchange_fn exchange, void *xarg);
/* new: the same with the library's own RCCL communicator (c3sc_hip_comm_*): a C main() shards over the GPUs of a node without
 * writing an exchange function.  id128 = the bytes of c3sc_hip_comm_unique_id from rank 0.  Collective.  0 on success. */
int c3control_shard_over_gpus(struct C3Control *, size_t world, size_t rank, const void *id128);
int c3control_comm_unique_id(void *id128); /* rank 0: the 128 bytes every rank passes to c3control_shard_over_gpus */
/* one value-iteration sweep's callback state, as c3control_step_vi builds it (bellman.c:2177-2199); the
 * cross approximation that consumes it (valuef_interp -> C3) is out of scope, so the caller drives the fibers */
struct VIparam *c3control_begin_vi(struct C3Control *, struct ValueF *vf, struct c3Opt *opt);
void c3control_end_vi(struct C3Control *, struct VIparam *, size_t *nevals);
/* the same for policy iteration: c3control_pi_solve's head (bellman.c:2351-2354) and c3control_step_pi's
 * callback state (bellman.c:2236-2249) */
struct PIparam *c3control_begin_pi(struct C3Control *, struct ValueF *policy);
void c3control_begin_pi_step(struct C3Control *, struct PIparam *, struct ValueF *vf, struct c3Opt *opt);
void c3control_end_pi_step(struct C3Control *, struct PIparam *, size_t *niter_evals);

/* ---- implicit policy for closed-loop simulation (bellman.c:2034-2042, 2105-2175) ---- */
void c3control_add_policy_sim(struct C3Control *, struct ValueF *pol, struct c3Opt *opt_sim,
                              void (*transform)(size_t, const double *, double *));
int c3control_policy_eval(struct C3Control *, double t, const double *x, double *u);
int c3control_controller(double t, const double *x, double *u, void *args /* struct C3Control * */);
/* new: Euler(-Maruyama) closed loop in place of the cdyn integrators the examples use; traj (nsteps+1) x dx */
int c3control_simulate(struct C3Control *, const double *x0, double dt, size_t nsteps, const double *noise, double *traj,
                       double *utraj);

/* ---- solver loops over the own cross driver (valuefunc.h: valuef_interp) ---- */
#include <stdio.h>
struct ApproxArgs;
struct Diag; /* bellman.c:2409-2514: per-iteration log */
void diag_destroy(struct Diag **head);
struct Diag *diag_create(size_t iter, int type, double norm, double abs_diff, size_t dim, size_t *ranks, double frac);
void diag_append(struct Diag **diag, size_t iter, int type, double norm, double abs_diff, size_t dim, size_t *ranks,
                 double frac);
void diag_print(struct Diag *head, FILE *fp);
int diag_save(struct Diag *head, char *filename);
size_t diag_count(const struct Diag *head);     /* new: read-only helpers for callers without the struct layout */
double diag_last_diff(const struct Diag *head);
struct ValueF *c3control_init_value(struct C3Control *, int (*f)(size_t, const double *, double *, void *), void *args,
                                    struct ApproxArgs *aargs, int verbose);                 /* bellman.c:2264-2280 */
struct ValueF *c3control_step_vi(struct C3Control *, struct ValueF *vf, struct ApproxArgs *, struct c3Opt *, int verbose,
                                 size_t *nevals);                                           /* bellman.c:2177-2212 */
struct ValueF *c3control_step_pi(struct C3Control *, struct ValueF *vf, struct PIparam *, struct ApproxArgs *,
                                 struct c3Opt *, int verbose, size_t *niter_evals);         /* bellman.c:2214-2262 */
struct ValueF *c3control_vi_solve(struct C3Control *, size_t maxiter, double abs_conv_tol, struct ValueF *vo,
                                  struct ApproxArgs *, struct c3Opt *, int verbose, struct Diag **diag); /* :2282-2340 */
struct ValueF *c3control_pi_solve(struct C3Control *, size_t maxiter, double abs_conv_tol, struct ValueF *policy,
                                  struct ApproxArgs *, struct c3Opt *, int verbose, struct Diag **diag); /* :2343-2407 */
#endif
