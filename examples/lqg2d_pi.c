/* lqg2d_pi.c -- a complete c3sc program against libc3sc.so: the call sequence of the reference's own regression
 * (test/transition_prob/tprob_test.c:2275-2364, Test_bellman_pi_100) and of its 2-D example main()s
 * (examples/lqg2d_new/lqg2d.c:255-383): set-up, initial value by interpolation, policy iteration interleaved with
 * value-iteration steps, save / reload, closed-loop simulation.  Own code; only the API names are the reference's.
 *
 *   cc -std=c99 -I include examples/lqg2d_pi.c -L c3sc_amd/host -lc3sc -lm -Wl,-rpath,$PWD/c3sc_amd/host -o lqg2d_pi
 *   ./lqg2d_pi [ngrid=60] [updates=5] [discount=0.1] [bruteforce|bfgs] [shard]
 *
 * "shard": the multi-GPU set-up of a C main() -- c3control_comm_unique_id + c3control_shard_over_gpus -- with WORLD_SIZE / RANK from
 * the environment (one process per GPU, device = C3SC_HIP_DEVICE; the 128-byte id travels through the file C3SC_COMM_ID_FILE).
 * Without a launcher it is a one-rank communicator: the same RCCL code path (librccl opened at run time) on one GPU.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "c3sc/c3sc.h"
#include "c3sc_hip.h" /* C3SC_MODEL_LQGND */

static int drift(double t, const double *x, const double *u, double *out, double *jac, void *a)
{
    (void)t; (void)a;
    out[0] = x[1];
    out[1] = u[0];
    if (jac) { jac[0] = 0.0; jac[1] = 1.0; }
    return 0;
}
static int diffusion(double t, const double *x, const double *u, double *out, double *grad, void *a)
{
    (void)t; (void)x; (void)u; (void)a;
    out[0] = 1.0; out[1] = 0.0; out[2] = 0.0; out[3] = 1.0;
    if (grad) memset(grad, 0, 4 * sizeof(double));
    return 0;
}
static int stagecost(double t, const double *x, const double *u, double *out, double *grad)
{
    (void)t;
    *out = x[0] * x[0] + x[1] * x[1] + u[0] * u[0];
    if (grad) grad[0] = 2.0 * u[0];
    return 0;
}
static int boundcost(double t, const double *x, double *out) { (void)t; (void)x; *out = 100.0; return 0; }
static int obscost(const double *x, double *out) { (void)x; *out = 0.0; return 0; }
static int startcost(size_t N, const double *x, double *out, void *arg)
{
    (void)x; (void)arg;
    for (size_t i = 0; i < N; i++) out[i] = 0.2;
    return 0;
}

int main(int argc, char **argv)
{
    size_t n = argc > 1 ? (size_t)atoi(argv[1]) : 60, updates = argc > 2 ? (size_t)atoi(argv[2]) : 5;
    double discount = argc > 3 ? atof(argv[3]) : 0.1;
    const int brute = !(argc > 4 && strcmp(argv[4], "bfgs") == 0);
    const int shard = argc > 5 && strcmp(argv[5], "shard") == 0;
    size_t dx = 2, du = 1, dw = 2, ngrid[2] = {n, n};
    double lb[2] = {-2.0, -2.0}, ub[2] = {2.0, 2.0};

    struct c3Opt *opt;
    if (brute) {
        double cands[33];
        for (int i = 0; i < 33; i++) cands[i] = -1.0 + 2.0 * i / 32.0;
        opt = c3opt_alloc(BRUTEFORCE, du);
        c3opt_set_brute_force_vals(opt, 33, cands);
    } else {
        double lbu = -1.0, ubu = 1.0;
        opt = c3opt_alloc(BFGS, du);
        c3opt_add_lb(opt, &lbu);
        c3opt_add_ub(opt, &ubu);
        c3opt_set_relftol(opt, 1e-8);
        c3opt_set_maxiter(opt, 10);
    }
    struct ApproxArgs *aargs = approx_args_init();
    approx_args_set_cross_tol(aargs, 1e-8);
    approx_args_set_round_tol(aargs, 1e-7);
    approx_args_set_kickrank(aargs, 5);
    approx_args_set_adapt(aargs, 1);
    approx_args_set_startrank(aargs, 5);
    approx_args_set_maxrank(aargs, 20);

    struct C3Control *c3c = c3control_create(dx, du, dw, lb, ub, ngrid, discount);
    c3control_add_drift(c3c, drift, NULL);
    c3control_add_diff(c3c, diffusion, NULL);
    c3control_add_stagecost(c3c, stagecost);
    c3control_add_boundcost(c3c, boundcost);
    c3control_add_obscost(c3c, obscost);
    c3control_set_external_boundary(c3c, 0, "reflect");
    c3control_set_external_boundary(c3c, 1, "reflect");
    const double prm[3] = {2.0, 1.0, 1.0}; /* the one line a maintainer adds: which device functor restates the callbacks */
    c3control_set_device_model(c3c, C3SC_MODEL_LQGND, prm, 3);

    if (shard) { /* one process per GPU: every rank runs this same program */
        const char *ws = getenv("WORLD_SIZE"), *rk = getenv("RANK"), *idf = getenv("C3SC_COMM_ID_FILE");
        const size_t world = ws ? (size_t)atoi(ws) : 1, rank = rk ? (size_t)atoi(rk) : 0;
        char id[128];
        if (rank == 0) {
            if (c3control_comm_unique_id(id) != 0) { fprintf(stderr, "c3control_comm_unique_id failed\n"); return 1; }
            if (world > 1 && idf) { FILE *fp = fopen(idf, "wb"); if (!fp || fwrite(id, 1, 128, fp) != 128) return 1; fclose(fp); }
        } else {
            FILE *fp = NULL;
            for (int tries = 0; tries < 600 && (fp = fopen(idf ? idf : "", "rb")) == NULL; tries++) { struct timespec ts = {0, 50000000}; nanosleep(&ts, NULL); }
            if (!fp || fread(id, 1, 128, fp) != 128) { fprintf(stderr, "rank %zu: no communicator id\n", rank); return 1; }
            fclose(fp);
        }
        if (c3control_shard_over_gpus(c3c, world, rank, id) != 0) return 1;
        printf("sharded over %zu rank(s) with the library's RCCL communicator\n", world);
    }
    struct ValueF *cost = c3control_init_value(c3c, startcost, NULL, aargs, 0);
    struct Diag *diag = NULL;
    double diff = 0.0;
    for (size_t ii = 0; ii < updates; ii++) {
        struct ValueF *next = c3control_pi_solve(c3c, 10, 1e-7, cost, aargs, opt, 0, &diag);
        valuef_destroy(cost);
        struct ValueF *temp = c3control_vi_solve(c3c, 1, 1e-7, next, aargs, opt, 0, &diag);
        diff = valuef_norm2diff(next, temp);
        cost = valuef_copy(temp);
        valuef_destroy(temp);
        valuef_destroy(next);
        printf("control update %zu: |V_vi - V_pi| = %.6e  |V| = %.6e  ranks 1 %zu 1\n", ii, diff, valuef_norm(cost), valuef_get_ranks(cost)[1]);
        if (diff < 1e-7) break;
    }
    diag_print(diag, stdout);

    char fname[] = "lqg2d_cost.c3sc";
    int ok = valuef_save(cost, fname) == 0;
    struct ValueF *back = valuef_load(fname, ngrid, c3control_get_xgrid(c3c));
    ok = ok && back != NULL && valuef_norm2diff(cost, back) <= 1e-12 * valuef_norm(cost);
    remove(fname);

    /* closed loop from (1.5, -1) with the implicit policy of the current value function (a few updates: not converged) */
    c3control_add_policy_sim(c3c, cost, opt, NULL);
    enum { NS = 400 };
    double x0[2] = {1.5, -1.0}, traj[(NS + 1) * 2], us[NS];
    ok = ok && c3control_simulate(c3c, x0, 0.01, NS, NULL, traj, us) == 0;
    const double r0 = hypot(x0[0], x0[1]), r1 = hypot(traj[2 * NS], traj[2 * NS + 1]);
    printf("closed loop: |x(0)| = %.3f -> |x(%.1f)| = %.3f, u(0) = %.3f\n", r0, 0.01 * NS, r1, us[0]);
    for (int i = 0; i <= NS; i++) ok = ok && isfinite(traj[2 * i]) && isfinite(traj[2 * i + 1]);
    for (int i = 0; i < NS; i++) ok = ok && fabs(us[i]) <= 1.0 + 1e-12;
    printf("%s\n", ok ? "LQG2D_PI_OK" : "LQG2D_PI_FAILED");

    valuef_destroy(back);
    valuef_destroy(cost);
    diag_destroy(&diag);
    c3control_destroy(c3c);
    c3opt_free(opt);
    approx_args_free(aargs);
    return ok ? 0 : 1;
}
