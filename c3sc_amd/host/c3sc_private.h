/* c3sc_private.h -- internals shared by the host sources of libc3sc.so (not installed) */
#ifndef C3SC_PRIVATE_H
#define C3SC_PRIVATE_H
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#define DIE(...)                                                                                   \
    do { fprintf(stderr, "c3sc: " __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)

void *c3sc_xcalloc(size_t n, size_t s);
#define xcalloc c3sc_xcalloc

struct c3sc_hip_ctx;
/* Nodal function train.  The reference keeps a C3 FunctionTrain + its evaluated cores + the cross index sets
 * (valuefunc.c:62-78); with linear elements the nodal core tables ARE the function train. */
struct ValueF {
    size_t d;
    size_t *N, *ranks;
    double **cores;              /* cores[m][j*r_m*r_{m+1} + a + b*r_m] (valuefunc.c:165-189) */
    double **grid;               /* nodes per dimension, or NULL (valuef_create_nodal without valuef_attach_grid) */
    size_t *nisl, *nisr;         /* cross index sets of the last interpolation (warm start), or NULL */
    int **isl, **isr;
    int sets_stable;             /* the interpolation that made this function ended with the index sets it started from (its predecessor's): the
                                    next one may try its first iteration speculatively (c3sc_hip_cross_speculate) */
    int elem_class;              /* 0 / LINELM: piecewise-linear between the nodes; CONSTELM: piecewise-constant (valuefunc.c:661-669) */
    unsigned long version;       /* bumps on every construction: identifies an upload */
    struct c3sc_hip_ctx *bound;  /* device context the cores were uploaded to */
};
void c3sc_forget_ctx(struct c3sc_hip_ctx *ctx);
/* valuef_interp_idx_sharded with a hook that is handed the rows the OTHER ranks computed (after the exchange): the solver's
 * value-iteration callback stores them in its node memo, so that every rank's memo has the content the unsharded run's
 * would have -- a node's value can depend on the direction of the fiber it was first computed in, and the
 * reference's memo keeps the first (bellman.c:1349-1353) */
#include <stdint.h>
typedef void (*c3sc_absorb_fn)(size_t F, size_t k, const int32_t *idx, const double *out, size_t lo, size_t hi, void *args);
struct ApproxArgs;
struct ValueF *c3sc_interp_idx_sharded(size_t d, int (*fi)(size_t, size_t, const int32_t *, double *, void *), void *args, const size_t *N,
                                       double **grid, struct ValueF *vref, struct ApproxArgs *aargs, int verbose, size_t world,
                                       size_t rank, int (*exchange)(double *, size_t, size_t, size_t, size_t, void *), void *xarg,
                                       c3sc_absorb_fn absorb);
struct ValueF *c3sc_interp_device(size_t d, struct c3sc_hip_ctx *ctx, int box, const size_t *N, double **grid, struct ValueF *vref,
                                  struct ApproxArgs *aargs, int verbose, size_t *nodes, struct c3sc_hip_ctx *policy_ctx, long long policy_tag,
                                  size_t *requested);
void valuef_set_cross_indices(struct ValueF *vf, const size_t *nisl, int *const *isl, const size_t *nisr, int *const *isr);
void valuef_free_cross_indices(struct ValueF *vf);
#endif
