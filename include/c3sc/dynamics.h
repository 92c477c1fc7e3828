/* dynamics.h -- holders of the user's drift / diffusion callbacks (mirrors src/dynamics.h:42-84). */
#ifndef C3SC_DYNAMICS_H
#define C3SC_DYNAMICS_H
#include <stddef.h>

struct Drift;
struct Diff;
struct Dyn;
typedef int (*c3sc_dyn_fn)(double, const double *, const double *, double *, double *, void *);

struct Drift *drift_alloc(size_t dx, size_t du);
struct Drift *drift_copy(struct Drift *);            /* dynamics.c:86-96: shares callback + argument */
void drift_free(struct Drift *);
void drift_add_func(struct Drift *, c3sc_dyn_fn, void *);
size_t drift_get_dx(struct Drift *);
size_t drift_get_du(struct Drift *);
int drift_eval(struct Drift *, double, const double *, const double *, double *, double *); /* dynamics.c:127-139 */

struct Diff *diff_alloc(size_t dx, size_t du, size_t dw);
struct Diff *diff_copy(struct Diff *);
void diff_free(struct Diff *);
void diff_add_func(struct Diff *, c3sc_dyn_fn, void *);
int diff_eval(struct Diff *, double, const double *, const double *, double *, double *);   /* dynamics.c:224-239 */
size_t diff_get_dw(struct Diff *);

/* dynamics.c:258-354: drift + diffusion pair.  dyn_alloc / dyn_init_ref borrow the two holders, the *_deep calls own them. */
struct Dyn *dyn_alloc(struct Drift *, struct Diff *);
struct Dyn *dyn_copy_deep(struct Dyn *);
void dyn_free(struct Dyn *);
void dyn_free_deep(struct Dyn *);
void dyn_init_ref(struct Dyn *, struct Drift *, struct Diff *);
size_t dyn_get_dx(struct Dyn *);
size_t dyn_get_dw(struct Dyn *);
size_t dyn_get_du(struct Dyn *);
int dyn_eval(struct Dyn *, double time, const double *x, const double *u, double *drift, double *jacdr, double *diff,
             double *jacdiff);
#endif
