/* valuefunc.h -- nodal function-train value function (mirrors src/valuefunc.h:47-76 for the hot path).
 * The reference keeps a C3 FunctionTrain inside struct ValueF; C3 is out of scope here, so the value
 * function is held directly as its nodal core tables (exactly what valuef_precompute_cores,
 * valuefunc.c:165-189, produces).  valuef_interp / save / load / norm live in C3. */
#ifndef C3SC_VALUE_H
#define C3SC_VALUE_H
#include <stddef.h>

struct ValueF;
/* new constructor: cores[m][j*r_m*r_{m+1} + a + b*r_m], ranks[d+1]; copies its inputs */
struct ValueF *valuef_create_nodal(size_t d, const size_t *N, const size_t *ranks, double **cores);
void valuef_destroy(struct ValueF *);
struct ValueF *valuef_copy(struct ValueF *);
size_t *valuef_get_ranks(struct ValueF *);
size_t valuef_get_dim(const struct ValueF *);
const size_t *valuef_get_N(const struct ValueF *);
double **valuef_get_cores(struct ValueF *);
/* value at a grid multi-index (what valuef_eval returns at grid nodes, valuefunc.c:345-350) */
double valuef_eval_ind(struct ValueF *, const size_t *ind);
/* valuefunc.c:369-585 -- runs on the GPU bound with valuef_bind_device */
int valuef_eval_fiber_ind_nn(struct ValueF *, const size_t *fixed_ind, size_t dim_vary, const size_t *neighbors,
                             const size_t *neighbors_vary, double *out);
struct c3sc_hip_ctx;
void valuef_bind_device(struct ValueF *, struct c3sc_hip_ctx *); /* uploads the cores (lazy, once per ctx) */
#endif
