/* The C ABI from plain C, no Python anywhere: fit 1000 local quadratics to samples of
 * f(x, y) = 1 + 2x - 3y + 0.5 x^2 + xy - 0.25 y^2 and check that every case recovers the exact derivatives at its origin.
 *
 *   gcc -O2 -I include examples/c/fit_quadratic.c -o fit_quadratic \
 *       -L python-wlsqm_amd/wlsqm/_lib -lwlsqm_hip -Wl,-rpath,$PWD/python-wlsqm_amd/wlsqm/_lib -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "wlsqm_hip.h"

static double f(double x, double y) { return 1.0 + 2.0 * x - 3.0 * y + 0.5 * x * x + x * y - 0.25 * y * y; }

int main(void) {
    enum { N = 1000, K = 20, NO = 6 };
    double* xk = malloc(sizeof(double) * N * K * 2); double* fk = malloc(sizeof(double) * N * K);
    double* xi = malloc(sizeof(double) * N * 2);     double* fi = calloc(N * NO, sizeof(double));
    int32_t* nk = malloc(sizeof(int32_t) * N); int32_t* order = malloc(sizeof(int32_t) * N); int32_t* wm = malloc(sizeof(int32_t) * N);
    int64_t* knowns = calloc(N, sizeof(int64_t));
    uint64_t s = 88172645463325252ull;
    for (int j = 0; j < N; ++j) {
        for (int m = 0; m < 2; ++m) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; xi[2 * j + m] = (double)(s >> 11) / 9007199254740992.0; }
        for (int k = 0; k < K; ++k) {
            for (int m = 0; m < 2; ++m) {
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                xk[(j * K + k) * 2 + m] = xi[2 * j + m] + 0.1 * ((double)(s >> 11) / 9007199254740992.0 - 0.5);
            }
            fk[j * K + k] = f(xk[(j * K + k) * 2], xk[(j * K + k) * 2 + 1]);
        }
        nk[j] = K; order[j] = 2; wm[j] = WLSQM_WEIGHT_CENTER;
    }
    if (wlsqm_hip_device_count() < 1) { fprintf(stderr, "no HIP device visible (libwlsqm_hip has no CPU fallback)\n"); return 2; }
    wlsqm_batch b = {0};
    b.dimension = 2; b.ncases = N; b.max_nk = K;
    b.xk = xk; b.xk_stride_case = K * 2; b.xk_stride_k = 2;
    b.fk = fk; b.fk_stride_case = K; b.fk_stride_k = 1;
    b.nk = nk; b.nk_stride = 1; b.xi = xi; b.xi_stride_case = 2; b.fi = fi; b.fi_stride_case = NO;
    b.order = order; b.order_stride = 1; b.knowns = knowns; b.knowns_stride = 1; b.weighting_method = wm; b.wm_stride = 1;
    int32_t iterations = -1;
    const int rc = wlsqm_hip_fit_many_host(&b, 0, &iterations);
    if (rc != WLSQM_OK) { fprintf(stderr, "fit failed (%d): %s\n", rc, wlsqm_hip_last_error()); return 1; }
    double worst = 0.0;
    for (int j = 0; j < N; ++j) {
        const double x = xi[2 * j], y = xi[2 * j + 1];
        const double want[NO] = {f(x, y), 2.0 + x + y, -3.0 + x - 0.5 * y, 1.0, 1.0, -0.5};   /* F, X, Y, X2, XY, Y2 */
        for (int a = 0; a < NO; ++a) worst = fmax(worst, fabs(fi[j * NO + a] - want[a]));
    }
    printf("%d cases, %d DOFs each (wlsqm_hip_number_of_dofs(2, 2) = %d): max |error| = %.3e, return value %d\n",
           N, NO, wlsqm_hip_number_of_dofs(2, 2), worst, (int)iterations);
    return worst < 1e-9 ? 0 : 1;
}
