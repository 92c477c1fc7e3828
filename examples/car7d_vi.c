/* car7d_vi.c -- value iteration to a tolerance on the headline configuration, as a plain C program against libc3sc.so:
 * the synthetic 7-D car (41^7 grid, FT rank 10, nine brute-force controls, discount 0), set up with the
 * reference's own calls (c3control_create / add_* / set_external_boundary / add_obstacle, bellman.c:1962-2103) and solved by
 * c3control_vi_solve's own loop and stopping test (bellman.c:2282-2340: stop when |V_i+1 - V_i|_L2 < abs_conv_tol).
 * Own code; only the API names are the reference's.
 *
 * Two knobs of this library's cross driver make the loop reach a 1e-3 step on this weakly contracting exit-time problem
 * (DESIGN.md 6.2): approx_args_set_crossrank -- the cross approximation of T(V) runs at ranks up to 48 and is cut back to the rank
 * cap by the TT-SVD, the value function (the kernels' input) keeps rank 10 -- and approx_args_set_cross_maxiter(1) -- one cross
 * iteration per sweep: the warm-started sweeps of a value iteration are the cross iterations.
 *
 *   cc -std=c99 -I include examples/car7d_vi.c -L c3sc_amd/host -lc3sc -lm -Wl,-rpath,$PWD/c3sc_amd/host -o car7d_vi
 *   ./car7d_vi [ngrid=41] [maxrank=10] [crossrank=48] [abs_tol=1.0] [maxiter=600]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "c3sc/c3sc.h"
#include "c3sc_hip.h" /* C3SC_MODEL_CAR7D */

#define PI 3.14159265358979323846

/* state (x, y, theta, v, omega, delta, a), controls (ddelta, da) */
static int drift(double t, const double *x, const double *u, double *out, double *jac, void *arg)
{
    (void)t; (void)arg;
    const double th = x[2], v = x[3], om = x[4], de = x[5], a = x[6];
    out[0] = v * cos(th);
    out[1] = v * sin(th);
    out[2] = om;
    out[3] = 2.0 * a;
    out[4] = (v / (0.2 * (1.0 + v / 8.0)) * tan(de) - om) / 0.5;
    out[5] = u[0];
    out[6] = u[1];
    if (jac) { memset(jac, 0, 14 * sizeof(double)); jac[5] = 1.0; jac[6 + 7] = 1.0; }
    return 0;
}
static int diffusion(double t, const double *x, const double *u, double *out, double *grad, void *arg)
{
    (void)t; (void)x; (void)u; (void)arg;
    memset(out, 0, 49 * sizeof(double));
    out[0] = 1.0; out[8] = 1.0;
    for (int i = 2; i < 7; i++) out[i * 7 + i] = 1e-2;
    if (grad) memset(grad, 0, 49 * 2 * sizeof(double));
    return 0;
}
static int stagecost(double t, const double *x, const double *u, double *out, double *grad)
{
    (void)t; (void)u;
    *out = 1.0 + pow(x[0], 2) + pow(x[1], 2);
    if (grad) { grad[0] = 0.0; grad[1] = 0.0; }
    return 0;
}
static int boundcost(double t, const double *x, double *out) { (void)t; (void)x; *out = 10.0; return 0; }
static int obscost(const double *x, double *out) { (void)x; *out = 0.0; return 0; }
static int startcost(size_t N, const double *x, double *out, void *arg)
{
    (void)x; (void)arg;
    for (size_t i = 0; i < N; i++) out[i] = 0.0;
    return 0;
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atoi(argv[1]) : 41, maxrank = argc > 2 ? (size_t)atoi(argv[2]) : 10;
    const size_t crossrank = argc > 3 ? (size_t)atoi(argv[3]) : 48;
    const double abs_tol = argc > 4 ? atof(argv[4]) : 1.0; /* ~1e-3 of |V|_L2 = 1.01e3 on the 41^7 grid */
    const size_t maxiter = argc > 5 ? (size_t)atoi(argv[5]) : 600;
    size_t dx = 7, du = 2, dw = 7, ngrid[7];
    for (int m = 0; m < 7; m++) ngrid[m] = n;
    double lb[7] = {-4.0, -4.0, -PI, 2.0, -2.0, -0.3, -1.0}, ub[7] = {4.0, 4.0, PI, 5.0, 2.0, 0.3, 1.0};

    double cands[18];
    const double c0[3] = {-0.5, 0.0, 0.5}, c1[3] = {-1.0, 0.0, 1.0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { cands[(i * 3 + j) * 2] = c0[i]; cands[(i * 3 + j) * 2 + 1] = c1[j]; }
    struct c3Opt *opt = c3opt_alloc(BRUTEFORCE, du);
    c3opt_set_brute_force_vals(opt, 9, cands);

    struct ApproxArgs *aargs = approx_args_init();
    approx_args_set_cross_tol(aargs, 1e-6);
    approx_args_set_round_tol(aargs, 1e-6);
    approx_args_set_kickrank(aargs, 4);
    approx_args_set_adapt(aargs, 1);
    approx_args_set_startrank(aargs, 4);
    approx_args_set_maxrank(aargs, maxrank);
    approx_args_set_crossrank(aargs, crossrank); /* new: cross approximation above the rank cap, rounded back to it */
    approx_args_set_cross_maxiter(aargs, 1);     /* new: one cross iteration per value-iteration sweep */

    struct C3Control *c3c = c3control_create(dx, du, dw, lb, ub, ngrid, 0.0);
    c3control_add_drift(c3c, drift, NULL);
    c3control_add_diff(c3c, diffusion, NULL);
    c3control_add_stagecost(c3c, stagecost);
    c3control_add_boundcost(c3c, boundcost);
    c3control_add_obscost(c3c, obscost);
    c3control_set_external_boundary(c3c, 0, "absorb");
    c3control_set_external_boundary(c3c, 1, "absorb");
    c3control_set_external_boundary(c3c, 2, "periodic");
    for (size_t m = 3; m < 7; m++) c3control_set_external_boundary(c3c, m, "reflect");
    double center[7] = {0.0, 0.0, 0.0, 3.5, 0.0, 0.0, 0.0}, width[7] = {1.0, 1.0, 2.0 * PI, 3.0, 4.0, 0.6, 2.0};
    c3control_add_obstacle(c3c, center, width); /* the goal box: cost 0 */
    c3control_set_device_model(c3c, C3SC_MODEL_CAR7D, NULL, 0); /* the one line a maintainer adds */

    struct ValueF *start = c3control_init_value(c3c, startcost, NULL, aargs, 0);
    struct Diag *diag = NULL;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    struct ValueF *cost = c3control_vi_solve(c3c, maxiter, abs_tol, start, aargs, opt, 0, &diag); /* bellman.c:2282-2340 */
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    const size_t sweeps = diag_count(diag);
    const double last = diag_last_diff(diag), norm = valuef_norm(cost);
    const size_t *ranks = valuef_get_ranks(cost);
    size_t rmax = 0;
    for (size_t m = 0; m <= dx; m++) if (ranks[m] > rmax) rmax = ranks[m];
    printf("c3control_vi_solve: %zu sweeps in %.2f s (%.1f ms per sweep); last |V_i+1 - V_i|_L2 = %.6e, |V|_L2 = %.6e (relative %.3e); "
           "FT rank of the value function %zu, cross rank %zu\n", sweeps, secs, 1e3 * secs / (double)(sweeps ? sweeps : 1), last, norm,
           last / norm, rmax, crossrank);
    const int converged = sweeps < maxiter && last < abs_tol;
    printf("%s\n", converged ? "CAR7D_VI_CONVERGED" : "CAR7D_VI_NOT_CONVERGED");
    const int ok = isfinite(norm) && norm > 0.0 && rmax <= maxrank;

    valuef_destroy(cost);
    valuef_destroy(start);
    diag_destroy(&diag);
    c3control_destroy(c3c);
    c3opt_free(opt);
    approx_args_free(aargs);
    return ok ? 0 : 1;
}
