/* c3sc_hip.h -- C-ABI of libc3sc_hip.so: the MI355X (gfx950) Bellman-backup engine.
 *
 * Plain C: opaque handle, plain pointers and sizes, int error codes.  No C++/torch types.
 * This is the boundary a c3sc maintainer binds from the C host code (INTEGRATION.md shows the
 * stub).  Every entry point names the reference interface it replaces; citations are
 * relative to the reference tree (goroda/c3sc).
 *
 * Data model
 *   grid      d per-dimension node arrays xgrid[m][0..N_m)     (c3control_create, bellman.c:1962-1999)
 *   boundary  EBTYPE per dim + <=10 box obstacles              (boundary.c:374-397, 470-481)
 *   mca       h2 = hmin^2, t[2m] = h2/h_m, t[2m+1] = h2/h_m^2  (mca_add_grid_refs, bellman.c:171-188)
 *   value     nodal FT cores, cores[m][j*r_m*r_{m+1} + a + b*r_m] (valuef_precompute_cores, valuefunc.c:165-189)
 *   controls  brute-force candidate list, U x du row-major      (c3opt_set_brute_force_vals, e.g. dubinscar.c:290-293)
 *   model     device functor id + params replacing the host drift/diff/stagecost/boundcost/obscost
 *             callbacks (dynamics.c:127-139,224-239; bellman.c:215-217), which a kernel cannot call.
 *   fibers    F grid fibers along dim k: int32 idx[F*d] fixed indices (entry k ignored)  -- the
 *             index form of the x[N x d] block C3's cross approximation hands to bellman_vi
 *             (bellman.c:1295; convert_fiber_to_ind nodeutil.c:437-470 does that mapping on the host).
 *
 * Pointers named d_* are DEVICE pointers (hipMalloc / torch CUDA tensors); h_* / unprefixed
 * configuration pointers are HOST pointers that are copied during the call.
 * `stream` is a hipStream_t passed as void* (NULL = default stream).
 */
#ifndef C3SC_HIP_H
#define C3SC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C3SC_MAX_DIM 12
#define C3SC_MAX_OBSTACLES 10 /* boundary.c:393 */
#define C3SC_MAX_PARAMS 8
#define C3SC_MAX_DU 4 /* control dimensions of the continuous (box) minimiser */

/* error codes (0 = ok, like the reference's int returns) */
enum {
    C3SC_OK = 0,
    C3SC_ERR_ARG = 1,         /* bad argument / state not set */
    C3SC_ERR_HIP = 2,         /* a HIP runtime call failed (c3sc_hip_last_error has the text) */
    C3SC_ERR_UNSUPPORTED = 3, /* no kernel instantiation for this (model, dim, rank, N) */
    C3SC_ERR_NODEVICE = 4
};

/* enum EBTYPE (boundary.h:42-47) */
enum { C3SC_EB_NONE = 0, C3SC_ABSORB = 1, C3SC_PERIODIC = 2, C3SC_REFLECT = 3 };

/* device problem models (restating the examples' callbacks; see c3sc_amd/csrc/models.hpp) */
enum {
    C3SC_MODEL_DUBINS3D = 1, /* examples/dubinscar_new/dubinscar.c:40-121 */
    C3SC_MODEL_SCAR4D = 2,   /* examples/skidding_car/scar.c:40-169 */
    C3SC_MODEL_CAR7D = 3,    /* synthetic 7-D car */
    C3SC_MODEL_LQGND = 4,    /* examples/lqgnd/lqgnd.c:80-198; params {dim, sig_even, sig_odd} */
    C3SC_MODEL_CHAIN = 5,    /* examples/double_int/double_int.c:80-157; params {dim, sig, sig_last, stage_mode} */
    C3SC_MODEL_ROSSLER3D = 6, /* examples/rossler/rossler.c:80-157; params {3, sig, sig_last} */
    C3SC_MODEL_PERCH7D = 8,  /* examples/perching/perch.c:36-273: glider perching, 7 states, elevator rate u in [-2 pi, 2 pi] */
    C3SC_MODEL_TPROB3D = 7,  /* the reference tests' 3-D problem: test/transition_prob/tprob_test.c f3 :223-251, s2, stagecost3d */
    C3SC_MODEL_SKID5D = 9,   /* examples/skidding5d/scar.c:39-176: 5-D skidding car (x, y, orientation, yaw rate, lateral speed), steering u */
    C3SC_MODEL_COTHRUST6D = 10, /* examples/cothrust2/copterposethrust.c:40-222: quadcopter position + velocity, controls (thrust, roll, pitch) */
    C3SC_MODEL_TABLE = 100   /* host-evaluated callbacks (c3sc_hip_bellman_fibers_tables); not set with set_model */
};

/* status bits accumulated by the kernels (c3sc_hip_get_status) */
enum {
    C3SC_STATUS_STATIONARY = 1u /* transition_assemble would have returned 1 (Q < 1e-14, nodeutil.c:365);
                                   the reference asserts (bellman.c:452); the candidate is skipped here */
};

/* kernel variants (c3sc_hip_set_variant); 0 lets the library choose.  Set the variant BEFORE uploading the value: the padded
 * rank of the device copy follows it (the quad kernel wants multiples of 4).  2 (one wavefront per 64 fibers) was retired (twice: DESIGN.md 4.5).
 * FIBER_QUAD covers both forms of that kernel (one or two wavefronts per 16 fibers).  Within a variant -- and across variants
 * under AUTO -- an instantiation whose LDS layout does not hold the grid declines and the next one of the same padded rank
 * runs; C3SC_ERR_UNSUPPORTED comes back only when none fits. */
enum { C3SC_VARIANT_AUTO = 0, C3SC_VARIANT_FIBER_PER_WAVE = 1, C3SC_VARIANT_FIBER_PER_LANE = 2, C3SC_VARIANT_FIBER_PAIR = 3,
       C3SC_VARIANT_FIBER_QUAD = 4 };

typedef struct c3sc_hip_ctx c3sc_hip_ctx;

/* lifetime: replaces c3control_create/destroy for the device side (bellman.c:1962-2019) */
int c3sc_hip_ctx_create(int device, c3sc_hip_ctx **out);
void c3sc_hip_ctx_destroy(c3sc_hip_ctx *ctx);
const char *c3sc_hip_last_error(const c3sc_hip_ctx *ctx);
int c3sc_hip_device_count(void);
/* largest FT rank the compiled kernels serve for (model id, state dimension d); 0 if none.  The reference has no such
 * limit (valuefunc.c:625-631 only clamps maxrank to min N); a caller clamps ApproxArgs.maxrank with it. */
int c3sc_hip_max_rank(int model, int d);

/* c3control_create's grid (bellman.c:1972-1986): ngrid[d], xgrid[m] host arrays of ngrid[m] doubles */
int c3sc_hip_set_grid(c3sc_hip_ctx *ctx, int d, const size_t *ngrid, const double *const *xgrid);
/* c3control_set_external_boundary / c3control_add_obstacle (bellman.c:2047-2062): bctype[d];
 * obstacles as inclusive boxes lb/ub (nobs x d row-major), lb = center - width/2 (boundary.c:264-267) */
int c3sc_hip_set_boundary(c3sc_hip_ctx *ctx, const int *bctype, int nobs, const double *obs_lb, const double *obs_ub);
/* NOT in the reference (default 0 = the reference's literal behaviour).  process_fibers_neighbor resets the absorbed flag of
 * a fiber's two end points from the varying dimension's own boundary type (nodeutil.c:570-612): a node on an absorbing face
 * of a fixed dimension, or inside an obstacle, is an ordinary node when it is the end point of a reflecting / periodic fiber
 * and a boundary / obstacle node along every other direction, so its value depends on the direction of the fiber it is
 * computed in (and, through the reference's memo, on which direction reached it first, bellman.c:1349-1353).  on = 1: end
 * points keep the flag the fixed dimensions and obstacles give them -- the batched Bellman operator becomes a function of
 * the node.  The solver loops of libc3sc.so switch it on (c3control_set_consistent_ends); the per-fiber entry points used
 * through the reference's callback ABI keep the literal behaviour. */
int c3sc_hip_set_consistent_ends(c3sc_hip_ctx *ctx, int on);
int c3sc_hip_get_consistent_ends(const c3sc_hip_ctx *ctx); /* 1 / 0; -1 for a null context */
/* mca_add_grid_refs (bellman.c:171-188) + dp_param_create's discount (bellman.c:220-235) */
int c3sc_hip_set_mca(c3sc_hip_ctx *ctx, double h2, const double *t, double discount);
/* replaces c3control_add_drift/diff/stagecost/boundcost/obscost (bellman.c:2064-2103) */
int c3sc_hip_set_model(c3sc_hip_ctx *ctx, int model, const double *params, int nparams);
/* replaces c3opt_alloc(BRUTEFORCE)+c3opt_set_brute_force_vals: cands[ncand*du], scanned in order, strict '<' */
int c3sc_hip_set_controls(c3sc_hip_ctx *ctx, int ncand, int du, const double *cands);
/* replaces vi_param_add_value + valuef_precompute_cores (bellman.c:1173, valuefunc.c:165-189):
 * ranks[d+1], cores[m] host arrays in the reference layout */
int c3sc_hip_upload_value(c3sc_hip_ctx *ctx, const size_t *ranks, const double *const *cores);
/* same, cores already resident on the device (e.g. after the RCCL all-gather of updated cores).  Asynchronous on `stream`: the
 * padded cores and the kernels' derived images are built by launches on that stream -- launch the fibers on the same stream or
 * synchronise first.  (c3sc_hip_upload_value above is complete on return.) */
int c3sc_hip_upload_value_device(c3sc_hip_ctx *ctx, const size_t *ranks, const double *const *d_cores, void *stream);
int c3sc_hip_set_variant(c3sc_hip_ctx *ctx, int variant);

/* THE HOT PATH.  Batched bellman_vi (bellman.c:1295-1423) without the memo: for each of the F
 * fibers along dim k and each of its N_k nodes: boundary stencil (process_fibers_neighbor,
 * nodeutil.c:489-627), FT neighbour costs (valuef_eval_fiber_ind_nn, valuefunc.c:369-585),
 * then bellman_optimal / bellman_control / transition_assemble / bellmanrhs per node
 * (bellman.c:504-543, 367-480; nodeutil.c:267-406; bellman.c:88-112).
 *   d_idx      int32 [F*d]        device
 *   d_out      double [F*N_k]     device, out[f*N_k + j] like the callback's out[]
 *   d_uidx     int32 [F*N_k] or NULL: winning candidate index, -1 for absorbed nodes
 *   d_absorbed int32 [F*N_k] or NULL: absorbed[] of process_fibers_neighbor (0 / 1 / -1)
 * Asynchronous on `stream`. */
int c3sc_hip_bellman_fibers(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *d_idx, double *d_out,
                            int32_t *d_uidx, int32_t *d_absorbed, void *stream);
/* The same for SEVERAL varying dimensions of one batch -- a sweep over independent fiber batches (SURVEY.md 8d's roofline batch, a
 * rank's share of it) -- as one call: segment s is dimension ks[s] with F[s] fibers, indices d_idx[s] (F[s] x d) and values
 * d_out[s] (F[s] x N_ks[s], distinct arrays); d_uidx / d_absorbed may be NULL or hold NULL entries.  The per-dimension launches
 * are independent, so the library spreads them over `stream` and two internal streams, forked from and joined to `stream` by
 * events: the next dimension's workgroups take the slots the previous dimension's last tiles leave, which d launches on one
 * stream cannot (car7d: 3 % off a sweep at 2^20 fibers per dimension, 8 % at 2^17; tools/multistream_probe.py).  Stream-ordered
 * like a single launch; results are those of the per-dimension calls, bit for bit (the same kernels).  C3SC_NO_OVERLAP=1 keeps
 * everything on `stream`. */
int c3sc_hip_bellman_fibers_all(c3sc_hip_ctx *ctx, int nk, const int *ks, const size_t *F, const int32_t *const *d_idx, double *const *d_out,
                                int32_t *const *d_uidx, int32_t *const *d_absorbed, void *stream);

/* Policy evaluation: batched bellman_pi (bellman.c:1702-1886) without its memo tables.  Same stencil and
 * neighbour costs from the uploaded value function (the reference's vf_iteration), but every node applies the
 * GIVEN control candidate instead of minimising: out = bellmanrhs(stage(u), discount, prob(u), dt(u), costs)
 * (:1807-1815, :1857-1865); absorbed / obstacle nodes get boundcost / obscost (:1787-1801).
 *   d_policy   int32 [F*N_k]  device: candidate index per node, as returned in d_uidx by
 *              c3sc_hip_bellman_fibers run on the policy's value function (-1 = no control: value 0)
 * The reference caches [prob, dt, stage] of the policy per node; here the candidate index is the cache and the
 * rates are recomputed (a few dozen flops). */
int c3sc_hip_policy_fibers(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *d_idx, const int32_t *d_policy,
                           double *d_out, int32_t *d_absorbed, void *stream);
/* ... and policy evaluation for several varying dimensions in one call (see c3sc_hip_bellman_fibers_all) */
int c3sc_hip_policy_fibers_all(c3sc_hip_ctx *ctx, int nk, const int *ks, const size_t *F, const int32_t *const *d_idx,
                               const int32_t *const *d_policy, double *const *d_out, int32_t *const *d_absorbed, void *stream);
int c3sc_hip_policy_fibers_host(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *h_idx, const int32_t *h_policy,
                                double *h_out, int32_t *h_absorbed);

/* Continuous controls (bellman_optimal's non-BRUTEFORCE branch, bellman.c:545-1118; the reference hands the node
 * objective to C3's BFGS with multistarts -- third party, unseeded for du >= 2, unpinned): the device minimises over
 * the box [lb, ub]^du with a tensor grid of `grid` points per control dimension followed by `polish` rounds of
 * coordinate golden-section search in the cell around the best grid point.  Models with per-candidate features
 * (C3SC_MODEL_SCAR4D: tan(u0)) are not served.  set_controls is not needed in this mode.
 *   d_uopt double [F*N_k*du] or NULL: the minimiser per node (zeros at absorbed nodes)
 * c3sc_hip_policy_fibers_box evaluates a given control per node (bellman_pi with continuous controls). */
int c3sc_hip_set_control_box(c3sc_hip_ctx *ctx, int du, const double *lb, const double *ub, int grid, int polish);
int c3sc_hip_bellman_fibers_box(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *d_idx, double *d_out, double *d_uopt,
                                int32_t *d_absorbed, void *stream);
int c3sc_hip_bellman_fibers_box_host(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *h_idx, double *h_out, double *h_uopt,
                                     int32_t *h_absorbed);
int c3sc_hip_policy_fibers_box(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *d_idx, const double *d_policy_u,
                               double *d_out, int32_t *d_absorbed, void *stream);
int c3sc_hip_policy_fibers_box_host(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *h_idx, const double *h_policy_u,
                                    double *h_out, int32_t *h_absorbed);

/* The same hot path for ARBITRARY host callbacks (the reference's examples unchanged): the host evaluates
 * drift_eval / diff_eval / stagecost (dynamics.c:127-139,224-239; bellman.c:414-444) for every (node, candidate)
 * and boundcost / obscost (bellman.c:458,467) for every node of the fibers it submits:
 *   d_tables double [F][N_k][U][2d+1] = (drift[d], diag(diffusion)[d], stage cost)
 *   d_costs2 double [F][N_k][2]       = (boundcost, obscost)
 * set_grid / set_boundary / set_mca / set_controls / upload_value must have been called; set_model is not needed. */
int c3sc_hip_bellman_fibers_tables(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *d_idx, const double *d_tables,
                                   const double *d_costs2, double *d_out, int32_t *d_uidx, int32_t *d_absorbed,
                                   void *stream);
int c3sc_hip_bellman_fibers_tables_host(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *h_idx,
                                        const double *h_tables, const double *h_costs2, double *h_out,
                                        int32_t *h_uidx, int32_t *h_absorbed);
/* policy evaluation (bellman_pi) with host-evaluated callbacks: as c3sc_hip_policy_fibers, rates from the tables */
int c3sc_hip_policy_fibers_tables(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *d_idx, const double *d_tables,
                                  const double *d_costs2, const int32_t *d_policy, double *d_out, int32_t *d_absorbed,
                                  void *stream);
int c3sc_hip_policy_fibers_tables_host(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *h_idx,
                                       const double *h_tables, const double *h_costs2, const int32_t *h_policy,
                                       double *h_out, int32_t *h_absorbed);

/* Batched mca_get_neighbor_costs (nodeutil.c:647-713) only: d_costs double [F*N_k*(2d+1)],
 * layout out[j*(2d+1) + 2m + {0,1}] = (-,+) neighbour in dim m, [.. + 2d] = self. */
int c3sc_hip_stencil_fibers(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *d_idx, double *d_costs,
                            int32_t *d_absorbed, void *stream);

/* The literal valuef_eval_fiber_ind_nn interface (valuefunc.c:369-371) batched: the caller supplies the
 * neighbour indices instead of having them derived from the boundary types.
 *   d_nb_fixed int32 [F][2(d-1)]  (-,+) neighbour index of every fixed dim, dims != k in order
 *   d_nb_vary  int32 [F][N_k][2]  (-,+) neighbour node of every fiber node
 * either may be NULL (then derived as in c3sc_hip_stencil_fibers). */
int c3sc_hip_stencil_fibers_nb(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *d_idx, const int32_t *d_nb_fixed,
                               const int32_t *d_nb_vary, double *d_costs, int32_t *d_absorbed, void *stream);
int c3sc_hip_stencil_fibers_nb_host(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *h_idx,
                                    const int32_t *h_nb_fixed, const int32_t *h_nb_vary, double *h_costs,
                                    int32_t *h_absorbed);

/* Convenience for host callers (the C facade's bellman_vi): host buffers, synchronous.  Batches whose buffers total
 * <= 1 MiB (a cross-approximation core step) are served from a pinned, device-mapped block of the context -- the kernel
 * reads the indices and writes its rows in place, no hipMemcpy; larger ones are staged through device scratch.
 * C3SC_NO_ZEROCOPY=1 in the environment forces the staged path. */
int c3sc_hip_bellman_fibers_host(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *h_idx, double *h_out,
                                 int32_t *h_uidx, int32_t *h_absorbed);
int c3sc_hip_stencil_fibers_host(c3sc_hip_ctx *ctx, int k, size_t F, const int32_t *h_idx, double *h_costs,
                                 int32_t *h_absorbed);

/* Device-resident core steps of the cross approximation that calls the path (valuefunc.c:603-767 hands bellman_vi to C3's
 * ftapprox_cross; c3sc_amd/host/c3sc_cross.c is this library's driver).  A cross iteration is 2 d sequential core steps of
 * r_k r_{k+1} fibers each; with fibers on the GPU and factorisation + node memo on the host a sweep is host-bound.  These
 * entry points keep a whole iteration on one stream: fiber index lists from the device-resident index sets, the batched
 * Bellman kernel, the node memo (first value stays: bellman.c:1333-1353, 1412-1417; keyed by node id and sweep epoch) and a
 * one-workgroup pivoted factorisation + maxvol that writes the interpolatory core and the next index set.
 *   ranks[d+1]; I[k]: ranks[k] tuples over dims 0..k-1 (int32, row-major); J[k]: ranks[k+1] tuples over dims k+1..d-1
 *   new_sweep != 0 starts a new memo epoch (workspace_increment_vi_iter, bellman.c:2199)
 *   box: 0 = candidate list (set_controls), 1 = control box (set_control_box)
 *   fetch waits for the stream and returns the cores of the last half sweep in the layout G_k[a + r_k (j + N_k b)], both
 *   families of index sets, and info[4] = {nodes stored in the memo since the last fetch (the reference's nnode_evals),
 *   1 if a fiber matrix was numerically rank deficient, maxvol row swaps, 1 if the memo overflowed / 2 if a rank of a sharded
 *   sweep failed (its rows arrived as NaN)}
 *   ranks up to 48 (above 32 a core step runs on global scratch).  Streams: setup works on the NULL stream and is COMPLETE when it
 *   returns, so iteration / confirm / speculate / fetch may be given any stream (all calls of one context on the same one) */
int c3sc_hip_cross_setup(c3sc_hip_ctx *ctx, const size_t *ranks, const int32_t *const *I, const int32_t *const *J, int new_sweep);
int c3sc_hip_cross_iteration(c3sc_hip_ctx *ctx, int box, void *stream);
/* The same iteration with its cores STREAMED to the host: the right-to-left half sweep produces the cores in the order k = d-1 .. 0,
 * the order in which a right-to-left orthogonalisation (the first half of the TT rounding, what C3's ftapprox_cross_rankadapt does
 * behind valuefunc.c:728-733) consumes them.  Each core is copied to the pinned block on a stream of its own as soon as its step
 * has run; c3sc_hip_cross_wait_core(k, h_core) waits for THAT copy only and hands the core over (working layout), while the later
 * steps are still running.  c3sc_hip_cross_fetch afterwards brings the index sets and counters (pass h_cores = NULL).  Same kernels,
 * same results as c3sc_hip_cross_iteration. */
int c3sc_hip_cross_iteration_streamed(c3sc_hip_ctx *ctx, int box, void *stream);
int c3sc_hip_cross_wait_core(c3sc_hip_ctx *ctx, int k, double *h_core);
/* after an iteration that changed index sets: the confirming iteration as ONE launch (all core steps side by side on the values
 * they already hold, comparing instead of writing their index sets).  *confirmed = 1: the iteration that would follow changes
 * nothing and its cores are in place -- fetch them; 0: run c3sc_hip_cross_iteration[_pi] as usual.  Synchronises the stream. */
int c3sc_hip_cross_confirm(c3sc_hip_ctx *ctx, int *confirmed, void *stream);
/* at the start of a sweep whose index sets come from the previous sweep: the whole iteration in d + 1 launches -- the fiber lists
 * of all d cores from the current sets, evaluated back to back (no core step in between), then the confirming launch above.
 * *confirmed = 1: every step reproduced its index set, so the sequential iteration would have returned exactly these cores (fetch
 * them); 0: run c3sc_hip_cross_iteration[_pi] -- what was evaluated here stays cached.  policy_ctx == NULL: bellman_vi's fibers, else
 * bellman_pi's as in c3sc_hip_cross_iteration_pi.  Unsharded contexts; otherwise nothing is launched and *confirmed = 0.  Because the
 * lists are evaluated before any set is known to survive, a failed attempt may have put nodes into the memo that the sequential
 * iteration would not have asked for: use it only where a node's value does not depend on the fiber that computes it
 * (c3sc_hip_set_consistent_ends).  Synchronises the stream. */
int c3sc_hip_cross_speculate(c3sc_hip_ctx *ctx, c3sc_hip_ctx *policy_ctx, long long policy_tag, int box, int *confirmed, void *stream);
/* the same for bellman_pi (bellman.c:1702-1886): per core step the greedy policy of the value function uploaded to policy_ctx
 * (cached per node for the whole policy iteration policy_tag: the reference's prob table, bellman.c:1806, 1877), then its
 * evaluation on ctx's value function.  info[0] of the fetch then counts the nodes whose policy was computed (npol_evals). */
int c3sc_hip_cross_iteration_pi(c3sc_hip_ctx *ctx, c3sc_hip_ctx *policy_ctx, long long policy_tag, void *stream);
/* pivot search of the core steps: warm_pivots != 0 starts it from the rows of the index set the step produced last time,
 * swap_tol is maxvol's dominance tolerance (row swaps while max |B| > 1 + swap_tol); defaults 1 and 0.05 */
int c3sc_hip_cross_options(c3sc_hip_ctx *ctx, int warm_pivots, double swap_tol);
/* after info[3] == 1 (memo full): double the memo tables keeping the current epoch's entries (no reference counterpart: the
 * reference's hash table never fills, util.c:760-766 -- it chains) */
int c3sc_hip_cross_grow_memo(c3sc_hip_ctx *ctx);
int c3sc_hip_cross_fetch(c3sc_hip_ctx *ctx, double *const *h_cores, int32_t *const *h_I, int32_t *const *h_J, unsigned long long *info,
                         void *stream);
void c3sc_hip_cross_free(c3sc_hip_ctx *ctx);

/* Multi-GPU (one process per GPU, RCCL over xGMI, opened at run time -- no link dependency).  The path shards
 * by independent fibers: every rank evaluates a contiguous block of each core step's fibers on its own device and ONE all-gather
 * per core step puts the F x N values on every rank (tens of KB, latency-bound).  bellman.c:2201 passes the value function
 * read-only during a sweep, which is what makes the fibers independent.
 *   unique_id: rank 0 creates the 128-byte id and hands it to the other ranks (file, environment, socket ...)
 *   create:    collective over all ranks of the node
 *   allgather: count doubles per rank, device buffers, in place when d_send == d_recv + rank * count; asynchronous on stream
 *   cross_set_comm: the device-resident cross iterations of ctx shard their core steps over the communicator
 *   exchange:  a c3sc_exchange_fn (include/c3sc/valuefunc.h) for the host-driven sharded driver: pass it with xarg = the
 *              communicator to c3control_set_fiber_sharding / valuef_interp_idx_sharded */
typedef struct c3sc_hip_comm c3sc_hip_comm;
int c3sc_hip_comm_unique_id(void *id128);
int c3sc_hip_comm_create(c3sc_hip_ctx *ctx, int world, int rank, const void *id128, c3sc_hip_comm **out);
void c3sc_hip_comm_destroy(c3sc_hip_comm *comm);
int c3sc_hip_comm_world(const c3sc_hip_comm *comm);
int c3sc_hip_comm_rank(const c3sc_hip_comm *comm);
int c3sc_hip_comm_allgather(c3sc_hip_comm *comm, const double *d_send, double *d_recv, size_t count, void *stream);
int c3sc_hip_cross_set_comm(c3sc_hip_ctx *ctx, c3sc_hip_comm *comm);
int c3sc_hip_comm_exchange(double *out, size_t F, size_t N, size_t lo, size_t hi, void *comm);

int c3sc_hip_sync(c3sc_hip_ctx *ctx, void *stream);
int c3sc_hip_get_status(c3sc_hip_ctx *ctx, unsigned *flags, int clear);
/* name of the kernel the last launch used (for profiles) */
const char *c3sc_hip_last_kernel(const c3sc_hip_ctx *ctx);

/* diagnostic builds only (C3SC_DBG & 128): per-wave segment cycle sums written by the kernels */
int c3sc_hip_debug_read(c3sc_hip_ctx *ctx, unsigned long long *out, size_t n);
/* diagnostics: Bellman / policy / stencil kernel launches made by this process so far (the reference calls its fiber
 * callback once per fiber, bellman.c:1295; here one launch serves a batch -- this counts them) */
unsigned long long c3sc_hip_launch_count(void);

/* device-side timing on `stream` with HIP events (used by bench.py's roofline leg) */
int c3sc_hip_timer_start(c3sc_hip_ctx *ctx, void *stream);
int c3sc_hip_timer_stop(c3sc_hip_ctx *ctx, void *stream, float *ms);

/* FP64 micro-benchmarks used to confirm the peaks quoted in DESIGN.md (results in TFLOP/s) */
int c3sc_hip_peak_fma_f64(c3sc_hip_ctx *ctx, double *tflops);
int c3sc_hip_peak_mfma_f64(c3sc_hip_ctx *ctx, double *tflops);

#ifdef __cplusplus
}
#endif
#endif
