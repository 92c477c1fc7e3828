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
    int elem_class;              /* 0 / LINELM: piecewise-linear between the nodes; CONSTELM: piecewise-constant (valuefunc.c:661-669) */
    unsigned long version;       /* bumps on every construction: identifies an upload */
    struct c3sc_hip_ctx *bound;  /* device context the cores were uploaded to */
};
void c3sc_forget_ctx(struct c3sc_hip_ctx *ctx);
void valuef_set_cross_indices(struct ValueF *vf, const size_t *nisl, int *const *isl, const size_t *nisr, int *const *isr);
void valuef_free_cross_indices(struct ValueF *vf);
#endif
