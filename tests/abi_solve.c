/* A C99 program that drives the hot path through the C-ABI only (no Python, no torch): a driven qubit
 *     y' = -i (w/2) X s(t) y,  s(t) = cos(2 pi nu t),
 * whose exact solution is y(T) = cos(theta) e0 - i sin(theta) e1 with theta = (w/2) int_0^T s = (w/2) sin(2 pi nu T) / (2 pi nu),
 * as a sweep of B instances with different drive strengths w_b -- fixed-step RK4 (midyn_rk4_solve) and the Magnus-2
 * propagator (midyn_expm_solve), a direct RHS evaluation (midyn_eval_rhs), and the one-rank RCCL broadcast of the
 * stack (midyn_comm_* / midyn_stack_broadcast in place, then midyn_stack_broadcast_from into a stack made by
 * midyn_stack_create_empty: the source is destroyed and EVERYTHING below runs on the received stack, i.e. on what a
 * non-root rank holds).  Built and run by tests/test_gpu_production_shapes.py with gcc;
 * libmidyn.so and the HIP runtime are dlopen'ed exactly as a non-Python host would do it.
 * Exit code 0 and a line "ABI_SOLVE_OK ..." on success. */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/midyn.h"

#define LOAD(name)                                              \
    *(void**)(&p_##name) = dlsym(h, #name);                     \
    if (!p_##name) {                                            \
        fprintf(stderr, "missing symbol %s\n", #name);          \
        return 2;                                               \
    }

static int (*p_midyn_ctx_create)(int, midyn_ctx**);
static int (*p_midyn_ctx_destroy)(midyn_ctx*);
static const char* (*p_midyn_last_error)(midyn_ctx*);
static int (*p_midyn_stack_create)(midyn_ctx*, int, int, const midyn_complex*, const midyn_complex*, const double*, void*,
                                   midyn_stack**);
static int (*p_midyn_stack_destroy)(midyn_stack*);
static int (*p_midyn_eval_rhs)(midyn_stack*, const double*, double, const midyn_complex*, int, midyn_complex*);
static int (*p_midyn_rk4_solve)(midyn_stack*, int, int, int, const double*, const double*, int, const int*, const double*,
                                const int*, int, const midyn_complex*, int, midyn_complex*);
static int (*p_midyn_expm_solve)(midyn_stack*, int, int, int, const double*, const double*, int, const int*, const double*,
                                 const int*, int, int, const midyn_complex*, int, midyn_complex*);
static int (*p_midyn_expm_plan_create)(midyn_stack*, int, int, int, const double*, int, const int*, const double*, const int*, int,
                                       int, const midyn_complex*, int, midyn_expm_plan**);
static int (*p_midyn_expm_plan_run)(midyn_expm_plan*, const double*, midyn_complex*);
static int (*p_midyn_expm_plan_fetch)(midyn_expm_plan*, midyn_complex*);
static int (*p_midyn_expm_plan_destroy)(midyn_expm_plan*);
static int (*p_midyn_expansion_create)(midyn_ctx*, int, int, const midyn_complex*, const midyn_complex*, const midyn_complex*, int,
                                       midyn_expansion**);
static int (*p_midyn_expansion_destroy)(midyn_expansion*);
static int (*p_midyn_expansion_solve)(midyn_expansion*, int, int, const double*, int, const midyn_complex*, int, midyn_complex*);
static int (*p_midyn_expansion_set_monomials)(midyn_expansion*, int, int, const int*);
static int (*p_midyn_expansion_solve_coeffs)(midyn_expansion*, int, int, const double*, int, const midyn_complex*, int, midyn_complex*);
static int (*p_midyn_comm_get_unique_id)(void*);
static int (*p_midyn_comm_init_rank)(midyn_ctx*, int, int, const void*, void**);
static int (*p_midyn_comm_destroy)(midyn_ctx*, void*);
static int (*p_midyn_stack_broadcast)(midyn_stack*, void*, int);
static int (*p_midyn_stack_broadcast_from)(midyn_stack*, midyn_stack*, void*, int);
static int (*p_midyn_stack_create_empty)(midyn_ctx*, int, int, int, int, midyn_stack**);

#define CHECK(ctx, call)                                                              \
    do {                                                                              \
        if ((call) != 0) {                                                            \
            fprintf(stderr, "%s failed: %s\n", #call, p_midyn_last_error(ctx));       \
            return 10;                                                                \
        }                                                                             \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 3) return 64;
    if (!dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL)) {   /* ONE HIP runtime, global, before libmidyn */
        fprintf(stderr, "hip runtime: %s\n", dlerror());
        return 3;
    }
    if (argc > 3 && !dlopen(argv[3], RTLD_NOW | RTLD_GLOBAL)) {   /* librccl next to that runtime (optional) */
        fprintf(stderr, "rccl: %s\n", dlerror());
        return 3;
    }
    void* h = dlopen(argv[2], RTLD_NOW);
    if (!h) {
        fprintf(stderr, "libmidyn: %s\n", dlerror());
        return 4;
    }
    LOAD(midyn_ctx_create) LOAD(midyn_ctx_destroy) LOAD(midyn_last_error) LOAD(midyn_stack_create)
    LOAD(midyn_stack_destroy) LOAD(midyn_eval_rhs) LOAD(midyn_rk4_solve) LOAD(midyn_expm_solve)
    LOAD(midyn_comm_get_unique_id) LOAD(midyn_comm_init_rank) LOAD(midyn_comm_destroy) LOAD(midyn_stack_broadcast)
    LOAD(midyn_stack_broadcast_from) LOAD(midyn_stack_create_empty)
    LOAD(midyn_expm_plan_create) LOAD(midyn_expm_plan_run) LOAD(midyn_expm_plan_fetch) LOAD(midyn_expm_plan_destroy)
    LOAD(midyn_expansion_create) LOAD(midyn_expansion_destroy) LOAD(midyn_expansion_solve) LOAD(midyn_expansion_set_monomials)
    LOAD(midyn_expansion_solve_coeffs)

    midyn_ctx* ctx = NULL;
    CHECK(NULL, p_midyn_ctx_create(0, &ctx));

    enum { N = 2, K = 1, B = 5, NSTEPS = 400 };
    const double T = 0.8, nu = 0.6, pi = 3.14159265358979323846;
    /* operator stack [k][n][n], already -i H: G_1 = -i X / 2 (the drive strength w_b is the per-instance coefficient) */
    midyn_complex ops[K * N * N] = {0.0, -0.5 * I, -0.5 * I, 0.0};
    midyn_stack* stack = NULL;
    CHECK(ctx, p_midyn_stack_create(ctx, N, K, ops, NULL, NULL, NULL, &stack));

    /* the one collective of the path, on a world of one rank */
    unsigned char id[MIDYN_COMM_ID_BYTES];
    void* comm = NULL;
    if (argc > 3) {
        CHECK(ctx, p_midyn_comm_get_unique_id(id));
        CHECK(ctx, p_midyn_comm_init_rank(ctx, 1, 0, id, &comm));
        CHECK(ctx, p_midyn_stack_broadcast(stack, comm, 0));
        /* the receiving side with data: an empty stack of the same shape is filled through ncclBroadcast */
        midyn_stack* received = NULL;
        CHECK(ctx, p_midyn_stack_create_empty(ctx, N, K, 0, 0, &received));
        CHECK(ctx, p_midyn_stack_broadcast_from(received, stack, comm, 0));
        CHECK(ctx, p_midyn_stack_destroy(stack));
        stack = received;
    }

    /* direct evaluation: G(t) y = c G_1 y */
    midyn_complex y[N] = {0.3 + 0.1 * I, -0.2 + 0.7 * I}, out[N];
    double c1 = 1.7;
    CHECK(ctx, p_midyn_eval_rhs(stack, &c1, 0.0, y, 1, out));
    if (cabs(out[0] - c1 * (-0.5 * I) * y[1]) > 1e-15 || cabs(out[1] - c1 * (-0.5 * I) * y[0]) > 1e-15) {
        fprintf(stderr, "eval_rhs mismatch\n");
        return 11;
    }

    /* RK4: distinct times t_s, t_s + h/2 (rows 2s, 2s+1) and the end point */
    const double hstep = T / NSTEPS;
    const int R = 2 * NSTEPS + 1;
    double* times = malloc(sizeof(double) * R);
    double* S = malloc(sizeof(double) * B * R * K);
    int* rows = malloc(sizeof(int) * 3 * NSTEPS);
    double* hs = malloc(sizeof(double) * NSTEPS);
    int* save = malloc(sizeof(int) * NSTEPS);
    double w[B];
    for (int r = 0; r < R; ++r) times[r] = 0.5 * hstep * r;
    for (int b = 0; b < B; ++b) {
        w[b] = 1.0 + 0.9 * b;
        for (int r = 0; r < R; ++r) S[(size_t)b * R + r] = w[b] * cos(2 * pi * nu * times[r]);
    }
    for (int s = 0; s < NSTEPS; ++s) {
        rows[3 * s] = 2 * s;
        rows[3 * s + 1] = 2 * s + 1;
        rows[3 * s + 2] = 2 * s + 2;
        hs[s] = hstep;
        save[s] = s == NSTEPS - 1 ? 1 : -1;
    }
    midyn_complex y0[N] = {1.0, 0.0};
    midyn_complex* Y = malloc(sizeof(midyn_complex) * B * 2 * N);
    CHECK(ctx, p_midyn_rk4_solve(stack, B, 1, R, times, S, NSTEPS, rows, hs, save, 2, y0, 1, Y));
    double worst_rk4 = 0.0;
    for (int b = 0; b < B; ++b) {
        const double theta = 0.5 * w[b] * sin(2 * pi * nu * T) / (2 * pi * nu);
        const midyn_complex e0 = cos(theta), e1 = -I * sin(theta);
        const midyn_complex* yb = Y + ((size_t)b * 2 + 1) * N;
        worst_rk4 = fmax(worst_rk4, fmax(cabs(yb[0] - e0), cabs(yb[1] - e1)));
        if (cabs(Y[(size_t)b * 2 * N] - 1.0) > 0.0) return 12;   /* slot 0 holds y0 */
    }

    /* Magnus-2 / expm with 40 steps: rows = the two Gauss points of every step */
    enum { ESTEPS = 40 };
    const double he = T / ESTEPS, g1 = 0.5 - sqrt(3.0) / 6, g2 = 0.5 + sqrt(3.0) / 6;
    const int RE = 2 * ESTEPS;
    double* te = malloc(sizeof(double) * RE);
    double* Se = malloc(sizeof(double) * B * RE);
    int rows_e[3 * ESTEPS], save_e[ESTEPS];
    double hse[ESTEPS];
    for (int s = 0; s < ESTEPS; ++s) {
        te[2 * s] = s * he + g1 * he;
        te[2 * s + 1] = s * he + g2 * he;
        rows_e[3 * s] = 2 * s;
        rows_e[3 * s + 1] = 2 * s + 1;
        rows_e[3 * s + 2] = 2 * s + 1;
        hse[s] = he;
        save_e[s] = s == ESTEPS - 1 ? 1 : -1;
    }
    for (int b = 0; b < B; ++b)
        for (int r = 0; r < RE; ++r) Se[(size_t)b * RE + r] = w[b] * cos(2 * pi * nu * te[r]);
    CHECK(ctx, p_midyn_expm_solve(stack, B, 1, RE, te, Se, ESTEPS, rows_e, hse, save_e, 2, 2, y0, 1, Y));
    double worst_expm = 0.0;
    for (int b = 0; b < B; ++b) {
        const double theta = 0.5 * w[b] * sin(2 * pi * nu * T) / (2 * pi * nu);
        const midyn_complex* yb = Y + ((size_t)b * 2 + 1) * N;
        worst_expm = fmax(worst_expm, fmax(cabs(yb[0] - cos(theta)), cabs(yb[1] + I * sin(theta))));
    }
    /* the same solve through the plan object (made once, run twice with different tables): the second run must equal
     * midyn_expm_solve of that table bit for bit, and the first the solve above */
    midyn_expm_plan* plan = NULL;
    midyn_complex* Yp = malloc(sizeof(midyn_complex) * B * 2 * N);
    midyn_complex* Yq = malloc(sizeof(midyn_complex) * B * 2 * N);
    CHECK(ctx, p_midyn_expm_plan_create(stack, B, 1, RE, te, ESTEPS, rows_e, hse, save_e, 2, 2, y0, 1, &plan));
    CHECK(ctx, p_midyn_expm_plan_run(plan, Se, NULL));
    CHECK(ctx, p_midyn_expm_plan_fetch(plan, Yp));
    if (memcmp(Yp, Y, sizeof(midyn_complex) * B * 2 * N) != 0) return 14;
    for (int i = 0; i < B * RE; ++i) Se[i] *= 0.5;
    CHECK(ctx, p_midyn_expm_plan_run(plan, Se, NULL));
    CHECK(ctx, p_midyn_expm_plan_fetch(plan, Yp));
    CHECK(ctx, p_midyn_expm_solve(stack, B, 1, RE, te, Se, ESTEPS, rows_e, hse, save_e, 2, 2, y0, 1, Yq));
    if (memcmp(Yp, Yq, sizeof(midyn_complex) * B * 2 * N) != 0) return 15;
    CHECK(ctx, p_midyn_expm_plan_destroy(plan));
    /* row f4: the array polynomial of a Dyson-type step, X_k = 1 + c0 A0 + c1 A1 + c0 c1 A01 (three terms, two coefficients), 6 steps,
     * two instances -- through the monomial table (midyn_expansion_solve) and through the coefficients + labels
     * (midyn_expansion_set_monomials / midyn_expansion_solve_coeffs): same bits, and the host's own product of the step matrices */
    {
        enum { XM = 3, XT = 6, XB = 2, XV = 2 };
        const midyn_complex terms[XM * N * N] = {0.0, -0.3 * I, -0.3 * I, 0.0, 0.2, 0.0, 0.0, -0.2, 0.05 * I, 0.01, -0.01, 0.02 * I};
        const midyn_complex ident[N * N] = {1.0, 0.0, 0.0, 1.0};
        const int labels[XM * 2] = {0, -1, 1, -1, 0, 1};
        double coeffs[XB * XV * XT], mono[XB * XT * XM];
        for (int b = 0; b < XB; ++b)
            for (int k = 0; k < XT; ++k) {
                const double c0 = 0.1 * (k + 1) * (b + 1), c1 = 0.3 - 0.05 * k + 0.02 * b;
                coeffs[(b * XV + 0) * XT + k] = c0;
                coeffs[(b * XV + 1) * XT + k] = c1;
                mono[(b * XT + k) * XM + 0] = c0;
                mono[(b * XT + k) * XM + 1] = c1;
                mono[(b * XT + k) * XM + 2] = c0 * c1;
            }
        midyn_expansion* ex = NULL;
        midyn_complex Ya[XB * N * N], Yb[XB * N * N];
        CHECK(ctx, p_midyn_expansion_create(ctx, N, XM, terms, ident, NULL, 0, &ex));
        CHECK(ctx, p_midyn_expansion_solve(ex, XB, XT, mono, N, ident, 1, Ya));
        CHECK(ctx, p_midyn_expansion_set_monomials(ex, XV, 2, labels));
        CHECK(ctx, p_midyn_expansion_solve_coeffs(ex, XB, XT, coeffs, N, ident, 1, Yb));
        if (memcmp(Ya, Yb, sizeof(Ya)) != 0) return 16;
        CHECK(ctx, p_midyn_expansion_solve_coeffs(ex, XB, XT, coeffs, N, ident, 1, Yb));     /* (the kept offset tables) */
        if (memcmp(Ya, Yb, sizeof(Ya)) != 0) return 17;
        for (int b = 0; b < XB; ++b) {
            midyn_complex P[N * N] = {1.0, 0.0, 0.0, 1.0};
            for (int k = 0; k < XT; ++k) {
                midyn_complex X[N * N], Q[N * N];
                for (int e = 0; e < N * N; ++e) {
                    X[e] = ident[e];
                    for (int i = 0; i < XM; ++i) X[e] += mono[(b * XT + k) * XM + i] * terms[i * N * N + e];
                }
                for (int r = 0; r < N; ++r)
                    for (int c = 0; c < N; ++c) {
                        Q[r * N + c] = 0.0;
                        for (int q = 0; q < N; ++q) Q[r * N + c] += X[r * N + q] * P[q * N + c];
                    }
                memcpy(P, Q, sizeof(P));
            }
            for (int e = 0; e < N * N; ++e)
                if (cabs(P[e] - Ya[b * N * N + e]) > 1e-13) return 18;
        }
        CHECK(ctx, p_midyn_expansion_destroy(ex));
    }
    if (comm) CHECK(ctx, p_midyn_comm_destroy(ctx, comm));
    CHECK(ctx, p_midyn_stack_destroy(stack));
    CHECK(ctx, p_midyn_ctx_destroy(ctx));
    printf("ABI_SOLVE_OK rk4_err=%.3e expm_err=%.3e broadcast=%d\n", worst_rk4, worst_expm, comm != NULL);
    /* RK4 with h = 2e-3: global error ~ h^4; Magnus-2 (4th order, the generators commute: exact up to the quadrature) */
    return (worst_rk4 < 1e-10 && worst_expm < 1e-8) ? 0 : 13;
}
