/*
 * c_driver.c -- libipm_hip.so from plain C, no Python and no torch: the library owns its workspace and stream
 * (workspace = NULL, stream = NULL in ipm_create).
 *
 * Solves the reference's small dense example (the comment optimum at main.py:1253 of payakorn/InteriorPointMethod,
 * -775): min -100 x1 - 125 x2 - 20 x3  s.t.  3 x1 + 6 x2 + 8 x3 = 30,  8 x1 + 4 x2 + x3 = 44,  x >= 0, with the
 * dense driver's conventions (start x = s = 1, y = 0; tol 1e-8), then a random strictly feasible 300 x 700 LP.
 *
 *   build:  cc -O2 -Iinclude examples/c_driver.c -o examples/c_driver -Linterior... -lipm_hip  (see __graft_entry__.build)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ipm_hip.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != IPM_OK) {                                                            \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ipm_last_error(h));           \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static double frand(unsigned long long* s) {   /* xorshift, uniform in (0,1) */
    *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
    return (double)((*s >> 11) + 1) / 9007199254740994.0;
}

static int solve_dense(const char* name, int m, int n, const double* A, const double* b, const double* c, double expect) {
    ipm_handle* h = NULL;
    ipm_stats st;
    CHECK(ipm_create(0, m, n, NULL, NULL, 0, NULL, &h));
    CHECK(ipm_set_A_dense(h, A, n, 0));
    CHECK(ipm_set_bc(h, b, c));
    CHECK(ipm_init_state(h, 0.0));
    CHECK(ipm_solve(h, 1e-8, 1e-8, 1e-8, 1000, &st));
    double* x = (double*)malloc(sizeof(double) * n);
    CHECK(ipm_get_state(h, x, NULL, NULL));
    double rp = st.rp_norm / (1.0 + st.b_norm), rd = st.rd_norm / (1.0 + st.c_norm);
    printf("%s: status %d iterations %d objective %.10f rp %.2e rd %.2e gap %.2e x[0] %.6f\n", name, st.status,
           st.iterations, st.objective, rp, rd, st.gap, x[0]);
    free(x);
    CHECK(ipm_destroy(h));
    if (st.status != IPM_STATUS_CONVERGED) return 1;
    if (!isnan(expect) && fabs(st.objective - expect) > 1e-6 * fmax(1.0, fabs(expect))) return 1;
    return 0;
}

int main(void) {
    int ndev = 0;
    if (ipm_device_count(&ndev) != IPM_OK || ndev < 1) { fprintf(stderr, "no HIP device: %s\n", ipm_last_error(NULL)); return 2; }
    printf("libipm_hip ABI %d, %d device(s)\n", ipm_abi_version(), ndev);

    const double A1[6] = {3, 6, 8, 8, 4, 1}, b1[2] = {30, 44}, c1[3] = {-100, -125, -20};
    if (solve_dense("ex1", 2, 3, A1, b1, c1, -775.0)) return 1;

    /* strictly feasible random LP: b = A x0, c = A^T y0 + s0 with x0, s0 > 0 */
    const int m = 300, n = 700;
    unsigned long long seed = 0x9E3779B97F4A7C15ull;
    double* A = (double*)malloc(sizeof(double) * m * n);
    double *x0 = (double*)malloc(sizeof(double) * n), *s0 = (double*)malloc(sizeof(double) * n);
    double *y0 = (double*)malloc(sizeof(double) * m), *b = (double*)calloc(m, sizeof(double)), *c = (double*)malloc(sizeof(double) * n);
    for (int i = 0; i < m * n; ++i) A[i] = 2.0 * frand(&seed) - 1.0;
    for (int j = 0; j < n; ++j) { x0[j] = 0.5 + frand(&seed); s0[j] = 0.5 + frand(&seed); c[j] = s0[j]; }
    for (int i = 0; i < m; ++i) y0[i] = 2.0 * frand(&seed) - 1.0;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) { b[i] += A[i * n + j] * x0[j]; c[j] += A[i * n + j] * y0[i]; }
    int rc = solve_dense("random 300x700", m, n, A, b, c, NAN);
    free(A); free(x0); free(s0); free(y0); free(b); free(c);
    return rc;
}
