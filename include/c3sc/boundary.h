/* boundary.h -- external boundary types, box obstacles and the BoundInfo query (mirrors src/boundary.h:42-87 of the
 * reference; the Bellman path reads boundary_type_dim / boundary_in_obstacle only). */
#ifndef C3SC_BOUNDARY_H
#define C3SC_BOUNDARY_H
#include <stddef.h>

enum EBTYPE { EB_NONE = 0, ABSORB = 1, PERIODIC = 2, REFLECT = 3 }; /* boundary.h:42-47 */

struct Boundary;
struct Boundary *boundary_alloc(size_t d, double *lb, double *ub);              /* boundary.c:374-397 */
struct Boundary *boundary_copy_deep(struct Boundary *old);                       /* boundary.c:402-424 */
void boundary_free(struct Boundary *b);                                          /* boundary.c:429-438 */
void boundary_external_set_type(struct Boundary *b, size_t dim, char *type);     /* "absorb" | "periodic" | "reflect" */
void boundary_add_obstacle(struct Boundary *b, double *center, double *lengths); /* boundary.c:470-481 */
size_t boundary_get_nobs(struct Boundary *b);
double *boundary_obstacle_get_lb(struct Boundary *b, size_t i);
double *boundary_obstacle_get_ub(struct Boundary *b, size_t i);
enum EBTYPE boundary_type_dim(const struct Boundary *b, size_t dim, int right);  /* boundary.c:604-614 */
/* boundary.c:577-597: x at/past a PERIODIC face -> the opposite face, *map = 1 (left->right) or 2 (right->left); else x, *map = 0 */
double outer_bound_dim(const struct Boundary *b, size_t dim, double x, int *map);
int boundary_in_obstacle(const struct Boundary *b, const double *x);             /* boundary.c:668-680 */
/* new (not in the reference; default off): end points of a reflecting / periodic fiber keep the absorbed flag the fixed
 * dimensions and obstacles give them (process_fibers_neighbor resets them, nodeutil.c:570-612) */
void boundary_set_consistent_ends(struct Boundary *b, int on);
int boundary_get_consistent_ends(const struct Boundary *b);
size_t boundary_get_dim(const struct Boundary *b);

/* where a state sits relative to the faces / obstacles (boundary.c:491-801) */
enum BOUNDRESULT { IN, LEFT, RIGHT };
struct BoundInfo;
struct BoundInfo *boundary_type(const struct Boundary *b, double time, const double *x);  /* caller frees: bound_info_free */
struct BoundInfo *bound_info_alloc(size_t d);
void bound_info_free(struct BoundInfo *);
int bound_info_set_dim(struct BoundInfo *, enum BOUNDRESULT, enum EBTYPE, size_t dim); /* 0 ok, 1 periodic (image owed), -1 unknown */
int bound_info_set_xmap_dim(struct BoundInfo *, double x, size_t dim);
int bound_info_onbound(const struct BoundInfo *);
int bound_info_onbound_dim(const struct BoundInfo *, size_t dim);
int bound_info_absorb(const struct BoundInfo *);
int bound_info_period(const struct BoundInfo *);
int bound_info_period_dim_dir(const struct BoundInfo *, size_t dim);   /* -1 left face, 1 right, 0 not periodic */
double bound_info_period_xmap(const struct BoundInfo *, size_t dim);
int bound_info_reflect(const struct BoundInfo *);
int bound_info_reflect_dim_dir(const struct BoundInfo *, size_t dim);
int bound_info_get_in_obstacle(const struct BoundInfo *);              /* obstacle index or -1 */
#endif
