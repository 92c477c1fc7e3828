/* nodeutil.h -- Markov-chain-approximation helpers (mirrors src/nodeutil.h:46-81, live functions only). */
#ifndef C3SC_NODEUTIL_H
#define C3SC_NODEUTIL_H
#include <stddef.h>

#include "boundary.h"
#include "valuefunc.h"

int transition_assemble(size_t dx, size_t du, size_t dw, double h, const double *hvec, const double *drift,
                        const double *grad_drift, const double *ddiff, const double *grad_ddiff, double *prob,
                        double *grad_prob, double *dt, double *grad_dt, double *space); /* nodeutil.c:267-406 */
/* the earlier form (nodeutil.c:82-233): h = minimum spacing, hvec[m] = spacing of dimension m */
int transition_assemble_old(size_t dx, size_t du, size_t dw, double h, const double *hvec, const double *drift,
                            const double *grad_drift, const double *ddiff, const double *grad_ddiff, double *prob,
                            double *grad_prob, double *dt, double *grad_dt, double *space);
int convert_fiber_to_ind(size_t d, size_t N, const double *x, const size_t *Ngrid, double **xgrid, size_t *fixed_ind,
                         size_t *dim_vary);                                             /* nodeutil.c:437-470 */
int process_fibers_neighbor(size_t d, const size_t *fixed_ind, size_t dim_vary, const double *x, int *absorbed,
                            size_t *neighbors_vary, size_t *neighbors_fixed, const size_t *ngrid,
                            const struct Boundary *bound);                              /* nodeutil.c:489-627 */
int mca_get_neighbor_costs(size_t d, size_t N, const double *x, struct Boundary *bound, struct ValueF *vf,
                           const size_t *ngrid, double **xgrid, size_t *fixed_ind, size_t *dim_vary, int *absorbed,
                           double *out);                                                /* nodeutil.c:647-713 */
/* nodeutil.c:718-816: stencil values around an OFF-GRID state (host; valuef_eval) */
int mca_get_neighbor_node_costs(size_t d, const double *x, struct Boundary *bound, struct ValueF *vf, const size_t *ngrid,
                                double **xgrid, int *absorbed, double *out);
#endif
