/* dynamics.h -- holders of the user's drift / diffusion callbacks (mirrors src/dynamics.h:46-71). */
#ifndef C3SC_DYNAMICS_H
#define C3SC_DYNAMICS_H
#include <stddef.h>

struct Drift;
struct Diff;
typedef int (*c3sc_dyn_fn)(double, const double *, const double *, double *, double *, void *);

struct Drift *drift_alloc(size_t dx, size_t du);
void drift_free(struct Drift *);
void drift_add_func(struct Drift *, c3sc_dyn_fn, void *);
size_t drift_get_dx(struct Drift *);
int drift_eval(struct Drift *, double, const double *, const double *, double *, double *); /* dynamics.c:127-139 */

struct Diff *diff_alloc(size_t dx, size_t du, size_t dw);
void diff_free(struct Diff *);
void diff_add_func(struct Diff *, c3sc_dyn_fn, void *);
int diff_eval(struct Diff *, double, const double *, const double *, double *, double *);   /* dynamics.c:224-239 */
size_t diff_get_dw(struct Diff *);
#endif
