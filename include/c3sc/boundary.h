/* boundary.h -- external boundary types and box obstacles (mirrors src/boundary.h:42-66 of the
 * reference for the part the Bellman path reads; the BoundInfo machinery, boundary.c:491-801, is
 * only used by the dead process_fibers and is out of scope). */
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
size_t boundary_get_dim(const struct Boundary *b);
#endif
