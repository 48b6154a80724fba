/* wlsqm_oracle.c — CPU restatement of the python-wlsqm batched fit path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under python-wlsqm_amd/ (the product) may
 * import, link, call or execute anything in oracle/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there
 * only as the checker / the reported CPU baseline.
 *
 * Parity status: PINNED.  This restatement is checked against golden vectors
 * captured from the real reference (Cython + OpenMP + LAPACK build of
 * /root/reference, see tests/golden/make_golden.py) in tests/test_oracle_golden.py:
 * fi / sens outputs, and the intermediates o2r, r2o, c, w, A, row_scale,
 * col_scale, ipiv.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Operation order follows the reference so that results
 * agree to rounding (no FMA contraction: build with -ffp-contract=off, as the
 * reference's -O2 x86-64 build has none; the only fused operations are the
 * explicit fma() calls the reference itself makes in polyeval.pyx).
 *
 * Third-party arithmetic on the path: LAPACK dgetrf/dgetrs via
 * scipy.linalg.cython_lapack (SciPy >= 1.9, unpinned; 1.15.3 + OpenBLAS 0.3.28
 * in the build container; call sites wlsqm/utils/lapackdrivers.pyx:1633,1663).
 * Not under /root/reference.  Restated here as the published unblocked
 * algorithm (LAPACK dgetf2: partial pivoting, first maximal |a_ik| wins,
 * column scaled by the reciprocal pivot; dgetrs('N'): row swaps, unit-lower
 * forward substitution, upper back substitution).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WEIGHT_UNIFORM 1   /* defs.pyx:74 */
#define WEIGHT_CENTER  2   /* defs.pyx:75 */

/* impl.pyx:30-31 */
static const double onesixth = 1. / 6.;
static const double one24th  = 1. / 24.;
/* infra.pyx:45-46 */
static const double weights_alpha = 1e-4;
static const double weights_beta  = 1. - 1e-4;
/* lapackdrivers.pyx:87 */
static const double ruiz_epsilon = 1e-15;

/* ---------------------------------------------------------------- helpers */

/* infra.pyx:67-112 */
int wlsqm_oracle_number_of_dofs(int dimension, int order) {
    static const int tab[3][5] = {
        {1, 2, 3, 4, 5},     /* defs.pyx:97-101  i1_*_end */
        {1, 3, 6, 10, 15},   /* defs.pyx:127-131 i2_*_end */
        {1, 4, 10, 20, 35}}; /* defs.pyx:177-181 i3_*_end */
    if (dimension < 1 || dimension > 3) return -1;
    if (order < 0 || order > 4) return -2;
    return tab[dimension - 1][order];
}

/* infra.pyx:119-121 (bits >= n are NOT masked off, as in the reference) */
int wlsqm_oracle_number_of_reduced_dofs(int n, long long mask) {
    return n - __builtin_popcountll((unsigned long long)mask);
}

/* infra.pyx:145-200 */
int wlsqm_oracle_remap(int* o2r, int* r2o, int n, long long mask) {
    int j, k = 0;
    for (j = 0; j < n; j++) {
        if (mask & (1LL << j)) o2r[j] = -1;
        else { o2r[j] = k; k++; }
    }
    for (j = 0; j < n; j++) {
        if (o2r[j] == -1) continue;
        r2o[o2r[j]] = j;
    }
    for (j = k; j < n; j++) r2o[j] = -1;
    return k;
}

/* ------------------------------------------------------------- case state */

typedef struct {
    int dimension, order, no, nr, nk, weighting_method;
    long long knowns;
    double xi, yi, zi;
    int *o2r, *r2o, *ipiv;
    double *c, *w, *A, *row_scale, *col_scale, *fi;
    double *wrk, *fk_tmp, *fi_tmp;
} Case; /* infra.pxd:124-182, minus the allocator plumbing */

/* infra.pyx:668-702 */
static void make_weights(Case* cs, double max_d2) {
    double* w = cs->w;
    int nk = cs->nk, k;
    if (cs->weighting_method == WEIGHT_UNIFORM) {
        for (k = 0; k < nk; k++) w[k] = 1.;
    } else { /* WEIGHT_CENTER and any other value, infra.pyx:691 */
        for (k = 0; k < nk; k++) {
            double d2 = w[k];
            double tmp = 1. - sqrt(d2 / max_d2);
            w[k] = weights_alpha + weights_beta * tmp * tmp;
        }
    }
}

/* impl.pyx:449-544; xk has element stride sk */
static void make_c_1D(Case* cs, const double* xk, long sk) {
    double* c = cs->c; double* w = cs->w;
    int order = cs->order, no = cs->no, nk = cs->nk, k;
    double xi = cs->xi, max_d2 = 0., dx, dx2;
    for (k = 0; k < nk; k++) {
        dx = xk[k * sk] - xi;
        dx2 = dx * dx;
        if (dx2 > max_d2) max_d2 = dx2;
        w[k] = dx2;
        c[k * no + 0] = 1.;
        if (order >= 1) c[k * no + 1] = dx;
        if (order >= 2) c[k * no + 2] = 0.5 * dx2;
        if (order >= 3) c[k * no + 3] = onesixth * dx * dx2;   /* impl.pyx:488,503 */
        if (order >= 4) c[k * no + 4] = one24th * dx2 * dx2;   /* impl.pyx:489 */
    }
    make_weights(cs, max_d2);
}

/* impl.pyx:286-432; xk[k*sk + {0,1}] */
static void make_c_2D(Case* cs, const double* xk, long sk) {
    double* c = cs->c; double* w = cs->w;
    int order = cs->order, no = cs->no, nk = cs->nk, k;
    double xi = cs->xi, yi = cs->yi, max_d2 = 0.;
    double dx, dy, dx2, dy2, dx3, dy3, d2;
    for (k = 0; k < nk; k++) {
        double* ck = c + (long)k * no;
        dx = xk[k * sk + 0] - xi;
        dy = xk[k * sk + 1] - yi;
        if (order >= 2) {
            dx2 = dx * dx; dy2 = dy * dy;
            d2 = dx2 + dy2;
        } else {
            dx2 = dy2 = 0.;
            d2 = dx * dx + dy * dy;                 /* impl.pyx:407,423 */
        }
        if (d2 > max_d2) max_d2 = d2;
        w[k] = d2;
        ck[0] = 1.;
        if (order >= 1) { ck[1] = dx; ck[2] = dy; }
        if (order >= 2) { ck[3] = 0.5 * dx2; ck[4] = dx * dy; ck[5] = 0.5 * dy2; }
        if (order == 3) {                            /* impl.pyx:376-379 */
            ck[6] = onesixth * dx2 * dx;
            ck[7] = 0.5 * dx2 * dy;
            ck[8] = 0.5 * dx * dy2;
            ck[9] = onesixth * dy * dy2;
        } else if (order == 4) {                     /* impl.pyx:319-349 */
            dx3 = dx2 * dx; dy3 = dy2 * dy;
            ck[6] = onesixth * dx3;
            ck[7] = 0.5 * dx2 * dy;
            ck[8] = 0.5 * dx * dy2;
            ck[9] = onesixth * dy3;
            ck[10] = one24th * dx2 * dx2;
            ck[11] = onesixth * dx3 * dy;
            ck[12] = 0.25 * dx2 * dy2;
            ck[13] = onesixth * dx * dy3;
            ck[14] = one24th * dy2 * dy2;
        }
    }
    make_weights(cs, max_d2);
}

/* impl.pyx:70-269; xk[k*sk + {0,1,2}].  DOF order defs.pyx:137-171. */
static void make_c_3D(Case* cs, const double* xk, long sk) {
    double* c = cs->c; double* w = cs->w;
    int order = cs->order, no = cs->no, nk = cs->nk, k;
    double xi = cs->xi, yi = cs->yi, zi = cs->zi, max_d2 = 0.;
    double dx, dy, dz, dx2, dy2, dz2, dx3, dy3, dz3, d2;
    for (k = 0; k < nk; k++) {
        double* ck = c + (long)k * no;
        dx = xk[k * sk + 0] - xi;
        dy = xk[k * sk + 1] - yi;
        dz = xk[k * sk + 2] - zi;
        if (order >= 2) {
            dx2 = dx * dx; dy2 = dy * dy; dz2 = dz * dz;
            d2 = dx2 + dy2 + dz2;
        } else {
            dx2 = dy2 = dz2 = 0.;
            d2 = dx * dx + dy * dy + dz * dz;       /* impl.pyx:240,260 */
        }
        if (d2 > max_d2) max_d2 = d2;
        w[k] = d2;
        ck[0] = 1.;
        if (order >= 1) { ck[1] = dx; ck[2] = dy; ck[3] = dz; }
        if (order >= 2) {
            ck[4] = 0.5 * dx2; ck[5] = dx * dy; ck[6] = 0.5 * dy2;
            ck[7] = dy * dz;   ck[8] = 0.5 * dz2; ck[9] = dx * dz;
        }
        if (order == 3) {                            /* impl.pyx:190-199 */
            ck[10] = onesixth * dx2 * dx;
            ck[11] = 0.5 * dx2 * dy;
            ck[12] = 0.5 * dx * dy2;
            ck[13] = onesixth * dy * dy2;
            ck[14] = 0.5 * dy2 * dz;
            ck[15] = 0.5 * dy * dz2;
            ck[16] = onesixth * dz * dz2;
            ck[17] = 0.5 * dx * dz2;
            ck[18] = 0.5 * dx2 * dz;
            ck[19] = dx * dy * dz;
        } else if (order == 4) {                     /* impl.pyx:102-157 */
            dx3 = dx2 * dx; dy3 = dy2 * dy; dz3 = dz2 * dz;
            ck[10] = onesixth * dx3;
            ck[11] = 0.5 * dx2 * dy;
            ck[12] = 0.5 * dx * dy2;
            ck[13] = onesixth * dy3;
            ck[14] = 0.5 * dy2 * dz;
            ck[15] = 0.5 * dy * dz2;
            ck[16] = onesixth * dz3;
            ck[17] = 0.5 * dx * dz2;
            ck[18] = 0.5 * dx2 * dz;
            ck[19] = dx * dy * dz;
            ck[20] = one24th * dx2 * dx2;
            ck[21] = onesixth * dx3 * dy;
            ck[22] = 0.25 * dx2 * dy2;
            ck[23] = onesixth * dx * dy3;
            ck[24] = one24th * dy2 * dy2;
            ck[25] = onesixth * dy3 * dz;
            ck[26] = 0.25 * dy2 * dz2;
            ck[27] = onesixth * dy * dz3;
            ck[28] = one24th * dz2 * dz2;
            ck[29] = onesixth * dx * dz3;
            ck[30] = 0.25 * dx2 * dz2;
            ck[31] = onesixth * dx3 * dz;
            ck[32] = 0.5 * dx2 * dy * dz;
            ck[33] = 0.5 * dx * dy2 * dz;
            ck[34] = 0.5 * dx * dy * dz2;
        }
    }
    make_weights(cs, max_d2);
}

/* impl.pyx:47-53 */
static void make_c_nD(Case* cs, const double* xk, long sk) {
    if (cs->dimension == 3) make_c_3D(cs, xk, sk);
    else if (cs->dimension == 2) make_c_2D(cs, xk, sk);
    else make_c_1D(cs, xk, sk);
}

/* impl.pyx:566-602: all nr*nr entries, k ascending, Fortran order */
static void make_A(Case* cs) {
    int nr = cs->nr, no = cs->no, nk = cs->nk, j, m, k;
    if (nr < 1) return;
    for (j = 0; j < nr; j++) {
        int oj = cs->r2o[j];
        for (m = 0; m < nr; m++) {
            int om = cs->r2o[m];
            double acc = 0.;
            for (k = 0; k < nk; k++)
                acc += cs->w[k] * cs->c[k * no + om] * cs->c[k * no + oj];
            cs->A[j + nr * m] = acc;
        }
    }
}

/* lapackdrivers.pyx:553-623 (+ init_scaling_c :285-290).  Returns iterations. */
static int rescale_ruiz2001(const double* A, int nrows, int ncols, double* row_scale, double* col_scale) {
    enum { NMAX = 35 };
    double DR[NMAX], DC[NMAX], DRprev[NMAX], DCprev[NMAX];
    int k, j, m;
    for (m = 0; m < ncols; m++) { col_scale[m] = 1.; DC[m] = 1.; DCprev[m] = 1.; }
    for (j = 0; j < nrows; j++) { row_scale[j] = 1.; DR[j] = 1.; DRprev[j] = 1.; }
    for (k = 0; k < 100; k++) {
        double acc, tmp;
        for (j = 0; j < nrows; j++) {
            double r = DRprev[j];
            acc = 0.;
            for (m = 0; m < ncols; m++) {
                tmp = fabs(A[j + nrows * m] / (r * DCprev[m]));
                if (tmp > acc) acc = tmp;
            }
            DR[j] = sqrt(acc);
        }
        for (m = 0; m < ncols; m++) {
            double cc = DCprev[m];
            acc = 0.;
            for (j = 0; j < nrows; j++) {
                tmp = fabs(A[j + nrows * m] / (cc * DRprev[j]));
                if (tmp > acc) acc = tmp;
            }
            DC[m] = sqrt(acc);
        }
        for (j = 0; j < nrows; j++) { DRprev[j] *= DR[j]; row_scale[j] /= DR[j]; }
        for (m = 0; m < ncols; m++) { DCprev[m] *= DC[m]; col_scale[m] /= DC[m]; }
        acc = fabs(1. - DR[0] * DR[0]);
        for (j = 1; j < nrows; j++) { tmp = fabs(1. - DR[j] * DR[j]); if (tmp > acc) acc = tmp; }
        if (acc < ruiz_epsilon) {
            acc = fabs(1. - DC[0] * DC[0]);
            for (m = 1; m < ncols; m++) { tmp = fabs(1. - DC[m] * DC[m]); if (tmp > acc) acc = tmp; }
            if (acc < ruiz_epsilon) break;
        }
    }
    return (k < 100) ? k + 1 : 100;
}

/* lapackdrivers.pyx:293-299 */
static void apply_scaling(double* A, int nrows, int ncols, const double* row_scale, const double* col_scale) {
    int j, m;
    for (m = 0; m < ncols; m++) {
        double cc = col_scale[m];
        for (j = 0; j < nrows; j++) A[j + nrows * m] *= (row_scale[j] * cc);
    }
}

/* dgetrf as called at lapackdrivers.pyx:1628-1635 (info ignored there).
 * Unblocked LAPACK dgetf2 semantics; ipiv is 1-based. */
static void lu_factor(double* A, int* ipiv, int n) {
    int j, i, m;
    for (j = 0; j < n; j++) {
        int p = j; double best = fabs(A[j + n * j]);
        for (i = j + 1; i < n; i++) {
            double v = fabs(A[i + n * j]);
            if (v > best) { best = v; p = i; }       /* first maximum wins (idamax) */
        }
        ipiv[j] = p + 1;
        if (A[p + n * j] != 0.) {
            if (p != j)
                for (m = 0; m < n; m++) { double t = A[j + n * m]; A[j + n * m] = A[p + n * m]; A[p + n * m] = t; }
            {
                double r = 1. / A[j + n * j];
                for (i = j + 1; i < n; i++) A[i + n * j] *= r;
            }
        }
        for (m = j + 1; m < n; m++) {
            double u = A[j + n * m];
            for (i = j + 1; i < n; i++) A[i + n * m] -= A[i + n * j] * u;
        }
    }
}

/* dgetrs('N') as called at lapackdrivers.pyx:1657-1665; one RHS, in place */
static void lu_solve(const double* LU, const int* ipiv, double* b, int n) {
    int i, j;
    for (i = 0; i < n; i++) {
        int p = ipiv[i] - 1;
        if (p != i) { double t = b[i]; b[i] = b[p]; b[p] = t; }
    }
    for (j = 0; j < n; j++)              /* L y = b, unit diagonal */
        for (i = j + 1; i < n; i++) b[i] -= LU[i + n * j] * b[j];
    for (j = n - 1; j >= 0; j--) {       /* U x = y */
        b[j] /= LU[j + n * j];
        for (i = 0; i < j; i++) b[i] -= LU[i + n * j] * b[j];
    }
}

/* impl.pyx:620-689 without the debug SVD */
static void preprocess_A(Case* cs) {
    if (cs->nr < 1) return;
    rescale_ruiz2001(cs->A, cs->nr, cs->nr, cs->row_scale, cs->col_scale);
    apply_scaling(cs->A, cs->nr, cs->nr, cs->row_scale, cs->col_scale);
    lu_factor(cs->A, cs->ipiv, cs->nr);
}

/* impl.pyx:731-846 (solve) and :861-974 (solve_contig): fk element stride sfk;
 * sens[k*ssk + n] (last axis contiguous) or NULL. */
static void solve(Case* cs, const double* fk, long sfk, double* fi, double* sens, long ssk, int do_sens) {
    int no = cs->no, nr = cs->nr, nk = cs->nk, j, k, om;
    const double *c = cs->c, *w = cs->w, *rs = cs->row_scale, *csc = cs->col_scale;
    const int* r2o = cs->r2o;
    double* b = cs->wrk;
    double* s = cs->wrk + nr;
    if (nr < 1) return;
    for (j = 0; j < nr; j++) {
        int oj = r2o[j];
        double acc = 0.;
        for (k = 0; k < nk; k++) {
            acc += w[k] * fk[k * sfk] * c[k * no + oj];
            if (do_sens) s[j + nr * k] = rs[j] * w[k] * c[k * no + oj];
        }
        b[j] = rs[j] * acc;
    }
    for (om = 0; om < no; om++) {
        if (cs->knowns & (1LL << om)) {
            for (j = 0; j < nr; j++) {
                int oj = r2o[j];
                for (k = 0; k < nk; k++)
                    b[j] -= fi[om] * w[k] * c[k * no + om] * c[k * no + oj] * rs[j];
            }
            if (do_sens)
                for (k = 0; k < nk; k++) sens[k * ssk + om] = NAN;
        }
    }
    lu_solve(cs->A, cs->ipiv, b, nr);
    if (do_sens)
        for (k = 0; k < nk; k++) lu_solve(cs->A, cs->ipiv, &s[k * nr], nr);
    for (j = 0; j < nr; j++) {
        int oj = r2o[j];
        fi[oj] = b[j] * csc[j];
        if (do_sens)
            for (k = 0; k < nk; k++) sens[k * ssk + oj] = s[nr * k + j] * csc[j];
    }
}

/* polyeval.pyx:874-951 */
static void taylor_1D(int order, const double* fi, double xi, const double* x, long sx, int n, double* out) {
    int k;
    for (k = 0; k < n; k++) {
        double dx = x[k * sx] - xi, acc;
        if (order == 4) {
            acc = fma(dx, one24th * fi[4], onesixth * fi[3]);
            acc = fma(dx, acc, 0.5 * fi[2]);
            acc = fma(dx, acc, fi[1]);
            out[k] = fma(dx, acc, fi[0]);
        } else if (order == 3) {
            acc = fma(dx, onesixth * fi[3], 0.5 * fi[2]);
            acc = fma(dx, acc, fi[1]);
            out[k] = fma(dx, acc, fi[0]);
        } else if (order == 2) {
            acc = fma(dx, 0.5 * fi[2], fi[1]);
            out[k] = fma(dx, acc, fi[0]);
        } else if (order == 1) {
            out[k] = fma(dx, fi[1], fi[0]);
        } else out[k] = fi[0];
    }
}

/* polyeval.pyx:550-735 */
static void taylor_2D(int order, const double* fi, double xi, double yi, const double* x, long sx, int n, double* out) {
    int k;
    for (k = 0; k < n; k++) {
        double dx = x[k * sx] - xi, dy = x[k * sx + 1] - yi, dxdy = dx * dy;
        double acc1, acc2, resX, resY, resXY;
        if (order == 4) {
            acc1 = fma(dy, fi[11], fi[6]);  acc1 *= onesixth; acc1 = fma(dx, one24th * fi[10], acc1);
            acc2 = fma(dy, fi[7], fi[3]);   acc2 *= 0.5;      acc2 = fma(dx, acc1, acc2);
            resX = fma(dx, acc2, fi[1]);
            acc1 = fma(dx, fi[13], fi[9]);  acc1 *= onesixth; acc1 = fma(dy, one24th * fi[14], acc1);
            acc2 = fma(dx, fi[8], fi[5]);   acc2 *= 0.5;      acc2 = fma(dy, acc1, acc2);
            resY = fma(dy, acc2, fi[2]);
            resXY = fma(dxdy, 0.25 * fi[12], fi[4]);
            acc1 = dxdy * resXY;
            acc1 = fma(dx, resX, acc1); acc1 = fma(dy, resY, acc1); acc1 += fi[0];
            out[k] = acc1;
        } else if (order == 3) {
            acc2 = fma(dy, fi[7], fi[3]); acc2 *= 0.5; acc2 = fma(dx, onesixth * fi[6], acc2);
            resX = fma(dx, acc2, fi[1]);
            acc2 = fma(dx, fi[8], fi[5]); acc2 *= 0.5; acc2 = fma(dy, onesixth * fi[9], acc2);
            resY = fma(dy, acc2, fi[2]);
            acc1 = dxdy * fi[4];
            acc1 = fma(dx, resX, acc1); acc1 = fma(dy, resY, acc1); acc1 += fi[0];
            out[k] = acc1;
        } else if (order == 2) {
            resX = fma(dx, 0.5 * fi[3], fi[1]);
            resY = fma(dy, 0.5 * fi[5], fi[2]);
            acc1 = dxdy * fi[4];
            acc1 = fma(dx, resX, acc1); acc1 = fma(dy, resY, acc1); acc1 += fi[0];
            out[k] = acc1;
        } else if (order == 1) {
            acc1 = dx * fi[1]; acc1 = fma(dy, fi[2], acc1); acc1 += fi[0];
            out[k] = acc1;
        } else out[k] = fi[0];
    }
}

/* polyeval.pyx:82-355 */
static void taylor_3D(int order, const double* fi, double xi, double yi, double zi, const double* x, long sx, int n, double* out) {
    int k;
    for (k = 0; k < n; k++) {
        double dx = x[k * sx] - xi, dy = x[k * sx + 1] - yi, dz = x[k * sx + 2] - zi;
        double dxdy = dx * dy, dydz = dy * dz, dxdz = dx * dz;
        double acc1, acc2, resX, resY, resZ, resXY, resYZ, resXZ;
        if (order == 4) {
            acc1 = fma(dy, fi[21], fi[10]); acc1 = fma(dz, fi[31], acc1); acc1 *= onesixth;
            acc1 = fma(dx, one24th * fi[20], acc1);
            acc2 = fma(dy, fi[11], fi[4]); acc2 = fma(dz, fi[18], acc2); acc2 = fma(dydz, fi[32], acc2);
            acc2 *= 0.5; acc2 = fma(dx, acc1, acc2);
            resX = fma(dx, acc2, fi[1]);
            acc1 = fma(dx, fi[23], fi[13]); acc1 = fma(dz, fi[25], acc1); acc1 *= onesixth;
            acc1 = fma(dy, one24th * fi[24], acc1);
            acc2 = fma(dx, fi[12], fi[6]); acc2 = fma(dz, fi[14], acc2); acc2 = fma(dxdz, fi[33], acc2);
            acc2 *= 0.5; acc2 = fma(dy, acc1, acc2);
            resY = fma(dy, acc2, fi[2]);
            acc1 = fma(dx, fi[29], fi[16]); acc1 = fma(dy, fi[27], acc1); acc1 *= onesixth;
            acc1 = fma(dz, one24th * fi[28], acc1);
            acc2 = fma(dx, fi[17], fi[8]); acc2 = fma(dy, fi[15], acc2); acc2 = fma(dxdy, fi[34], acc2);
            acc2 *= 0.5; acc2 = fma(dz, acc1, acc2);
            resZ = fma(dz, acc2, fi[3]);
            resXY = fma(dxdy, 0.25 * fi[22], fi[5]);
            resYZ = fma(dydz, 0.25 * fi[26], fi[7]);
            resXZ = fma(dxdz, 0.25 * fi[30], fi[9]);
            acc1 = dx * dy * dz * fi[19];
            acc1 = fma(dxdy, resXY, acc1); acc1 = fma(dydz, resYZ, acc1); acc1 = fma(dxdz, resXZ, acc1);
            acc1 = fma(dx, resX, acc1); acc1 = fma(dy, resY, acc1); acc1 = fma(dz, resZ, acc1);
            acc1 += fi[0];
            out[k] = acc1;
        } else if (order == 3) {
            acc2 = fma(dy, fi[11], fi[4]); acc2 = fma(dz, fi[18], acc2); acc2 *= 0.5;
            acc2 = fma(dx, onesixth * fi[10], acc2);
            resX = fma(dx, acc2, fi[1]);
            acc2 = fma(dx, fi[12], fi[6]); acc2 = fma(dz, fi[14], acc2); acc2 *= 0.5;
            acc2 = fma(dy, onesixth * fi[13], acc2);
            resY = fma(dy, acc2, fi[2]);
            acc2 = fma(dx, fi[17], fi[8]); acc2 = fma(dy, fi[15], acc2); acc2 *= 0.5;
            acc2 = fma(dz, onesixth * fi[16], acc2);
            resZ = fma(dz, acc2, fi[3]);
            acc1 = dx * dy * dz * fi[19];
            acc1 = fma(dxdy, fi[5], acc1); acc1 = fma(dydz, fi[7], acc1); acc1 = fma(dxdz, fi[9], acc1);
            acc1 = fma(dx, resX, acc1); acc1 = fma(dy, resY, acc1); acc1 = fma(dz, resZ, acc1);
            acc1 += fi[0];
            out[k] = acc1;
        } else if (order == 2) {
            resX = fma(dx, 0.5 * fi[4], fi[1]);
            resY = fma(dy, 0.5 * fi[6], fi[2]);
            resZ = fma(dz, 0.5 * fi[8], fi[3]);
            acc1 = dxdy * fi[5];
            acc1 = fma(dydz, fi[7], acc1); acc1 = fma(dxdz, fi[9], acc1);
            acc1 = fma(dx, resX, acc1); acc1 = fma(dy, resY, acc1); acc1 = fma(dz, resZ, acc1);
            acc1 += fi[0];
            out[k] = acc1;
        } else if (order == 1) {
            acc1 = dx * fi[1]; acc1 = fma(dy, fi[2], acc1); acc1 = fma(dz, fi[3], acc1); acc1 += fi[0];
            out[k] = acc1;
        } else out[k] = fi[0];
    }
}

/* impl.pyx:986-1083; model evaluation via interp.pyx:252-258 with diff=0 -> taylor_*D
 * (interp.pyx:281 3D, and the corresponding diff==0 branches for 2D/1D). */
static int solve_iterative(Case* cs, const double* fk, long sfk, double* sens, long ssk, int do_sens,
                           int max_iter, const double* xk, long sk) {
    int nk = cs->nk, no = cs->no, om, i = 0, k, broke = 0;
    double* wrk_fk = cs->fk_tmp; double* wrk_fi = cs->fi_tmp; double* fi = cs->fi;
    double norm, prev_norm = -1., tmp;
    solve(cs, fk, sfk, fi, sens, ssk, do_sens);
    /* The reference zeroes the entries of the KNOWN DOFs only (impl.pyx:1004-1008).  With stray mask bits beyond `no` the unknown that
     * drops out of the reduced system (infra.pyx:119-121) is never written by the correction solves either, and fi[om] += wrk_fi[om]
     * (impl.pyx:1076-1078) then adds an UNINITIALISED entry of the work array in every sweep: undefined in the reference (whatever the
     * heap held; a fresh heap holds zeros).  The oracle defines it as 0 — the fresh-heap behaviour, and what the GPU kernels do (the
     * dropped DOF keeps the caller's value) — so that its output does not depend on the allocator's history. */
    for (om = 0; om < no; om++) wrk_fi[om] = 0.;
    for (om = 0; om < no; om++)
        if (cs->knowns & (1LL << om)) wrk_fi[om] = 0.;
    for (i = 0; i < max_iter; i++) {
        if (cs->dimension == 3) taylor_3D(cs->order, fi, cs->xi, cs->yi, cs->zi, xk, sk, nk, wrk_fk);
        else if (cs->dimension == 2) taylor_2D(cs->order, fi, cs->xi, cs->yi, xk, sk, nk, wrk_fk);
        else taylor_1D(cs->order, fi, cs->xi, xk, sk, nk, wrk_fk);
        for (k = 0; k < nk; k++) wrk_fk[k] = fk[k * sfk] - wrk_fk[k];
        norm = fabs(wrk_fk[0]);
        for (k = 1; k < nk; k++) { tmp = fabs(wrk_fk[k]); if (tmp > norm) norm = tmp; }
        if (norm == prev_norm) { broke = 1; break; }
        prev_norm = norm;
        solve(cs, wrk_fk, 1, wrk_fi, NULL, 0, 0);
        for (om = 0; om < no; om++)
            if (!(cs->knowns & (1LL << om))) fi[om] += wrk_fi[om];
    }
    if (!broke) i = (max_iter > 0) ? max_iter : 1;   /* for/else (impl.pyx:1080-1081): i += 1 after the last pass;
                                                        with max_iter <= 0 the body never runs and i = 0 + 1 */
    return i;
}

/* ------------------------------------------------------------ batch driver */

static size_t case_bytes(int no, int nr, int nk, int do_sens) {
    size_t d = (size_t)nk * no + nk + (size_t)nr * nr + 2 * nr + no /*fi*/
             + (do_sens ? (size_t)nr * (nk + 1) : (size_t)nr) + nk + no;
    return d * sizeof(double) + (size_t)(2 * no + nr) * sizeof(int) + 64;
}

static void case_bind(Case* cs, char* buf, int do_sens) {
    int no = cs->no, nr = cs->nr, nk = cs->nk;
    double* d = (double*)buf;
    cs->c = d; d += (size_t)nk * no;
    cs->w = d; d += nk;
    cs->A = d; d += (size_t)(nr > 0 ? nr * nr : 0);
    cs->row_scale = d; d += (nr > 0 ? nr : 0);
    cs->col_scale = d; d += (nr > 0 ? nr : 0);
    cs->fi = d; d += no;
    cs->wrk = d; d += (nr > 0 ? (do_sens ? (size_t)nr * (nk + 1) : (size_t)nr) : 0);
    cs->fk_tmp = d; d += nk;
    cs->fi_tmp = d; d += no;
    cs->o2r = (int*)d; cs->r2o = cs->o2r + no; cs->ipiv = cs->r2o + no;
}

/* Optional capture of per-case intermediates for golden pinning (any pointer may be NULL).
 * Layouts: o2r,r2o [ncases,35] int32; c [ncases,max_nk,35]; w [ncases,max_nk];
 * A_unscaled, LU [ncases,35*35] (Fortran nr x nr packed at the front);
 * row_scale, col_scale [ncases,35]; ipiv [ncases,35] (1-based). */
typedef struct {
    int* o2r; int* r2o; double* c; double* w; double* A; double* row_scale; double* col_scale;
    double* LU; int* ipiv; long max_nk;
} OracleDebug;

/* One case, start to finish: simple.pyx:996-1008 (loop body of generic_fit_basic_many_parallel),
 * = make_c_nD, make_A, preprocess_A, Case_set_fi, solve / solve_iterative.
 * The result is left in cs->fi; the caller commits it (simple.pyx:1018-1019). */
static int fit_one(Case* cs, const double* xk, long sk, const double* fk, long sfk, const double* fi_user,
                   double* sens, long ssk, int do_sens, int iterative, int max_iter, OracleDebug* dbg, long j) {
    int om, it = 0;
    wlsqm_oracle_remap(cs->o2r, cs->r2o, cs->no, cs->knowns);   /* infra.pyx:872 (Case_allocate) */
    make_c_nD(cs, xk, sk);
    make_A(cs);
    if (dbg) {
        if (dbg->o2r) memcpy(dbg->o2r + j * 35, cs->o2r, cs->no * sizeof(int));
        if (dbg->r2o) memcpy(dbg->r2o + j * 35, cs->r2o, cs->no * sizeof(int));
        if (dbg->c) { int kk; for (kk = 0; kk < cs->nk; kk++)
            memcpy(dbg->c + (j * dbg->max_nk + kk) * 35, cs->c + (size_t)kk * cs->no, cs->no * sizeof(double)); }
        if (dbg->w) memcpy(dbg->w + j * dbg->max_nk, cs->w, cs->nk * sizeof(double));
        if (dbg->A && cs->nr > 0) memcpy(dbg->A + j * 1225, cs->A, (size_t)cs->nr * cs->nr * sizeof(double));
    }
    preprocess_A(cs);
    if (dbg && cs->nr > 0) {
        if (dbg->row_scale) memcpy(dbg->row_scale + j * 35, cs->row_scale, cs->nr * sizeof(double));
        if (dbg->col_scale) memcpy(dbg->col_scale + j * 35, cs->col_scale, cs->nr * sizeof(double));
        if (dbg->LU) memcpy(dbg->LU + j * 1225, cs->A, (size_t)cs->nr * cs->nr * sizeof(double));
        if (dbg->ipiv) memcpy(dbg->ipiv + j * 35, cs->ipiv, cs->nr * sizeof(int));
    }
    for (om = 0; om < cs->no; om++) cs->fi[om] = fi_user[om];   /* Case_set_fi, infra.pyx:780-785 */
    if (iterative) it = solve_iterative(cs, fk, sfk, sens, ssk, do_sens, max_iter, xk, sk);
    else solve(cs, fk, sfk, cs->fi, sens, ssk, do_sens);
    return it;
}

/* Batch fit: restates simple.pyx:953-1058 (basic) and :1065-1170 (iterative), and with
 * ntasks==1 the serial drivers :731-831 / :850-942.  All strides are in ELEMENTS.
 *   xk  [j*sxk_j + k*sxk_k + m]   (dimension 1: m absent)
 *   fk  [j*sfk_j + k*sfk_k]
 *   xi  [j*sxi_j + m]
 *   fi  [j*sfi_j + n]        in/out
 *   sens[j*ss_j + k*ss_k + n] or NULL
 *   nk, order, knowns, wm: per-case arrays with element strides.
 * Returns max refinement iterations (0 for basic), or <0 on error. */
int wlsqm_oracle_fit_many(int dimension, long ncases,
                          const double* xk, long sxk_j, long sxk_k,
                          const double* fk, long sfk_j, long sfk_k,
                          const int* nk, long snk,
                          const double* xi, long sxi_j,
                          double* fi, long sfi_j,
                          double* sens, long ss_j, long ss_k, int do_sens,
                          const int* order, long sorder,
                          const long long* knowns, long sknowns,
                          const int* wm, long swm,
                          int iterative, int max_iter, int ntasks, OracleDebug* dbg) {
    long j;
    int max_it = 0;
    int bad = 0;
    if (dimension < 1 || dimension > 3) return -1;
    if (ntasks < 1) return -3;
    if (ncases < 1) return -4;   /* infra.pyx:311-313 max_cases < 1 -> ValueError */

    /* Results are staged and committed after ALL solves (aliasing guarantee, simple.pyx:1010-1019). */
    double* fi_stage = (double*)malloc((size_t)ncases * 35 * sizeof(double));
    int* no_arr = (int*)malloc((size_t)ncases * sizeof(int));
    if (!fi_stage || !no_arr) { free(fi_stage); free(no_arr); return -5; }

#ifdef _OPENMP
#pragma omp parallel num_threads(ntasks) reduction(max : max_it) reduction(| : bad)
#endif
    {
        size_t cap = 0; char* buf = NULL;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (j = 0; j < ncases; j++) {
            Case cs;
            cs.dimension = dimension;
            cs.order = order[j * sorder];
            cs.knowns = knowns[j * sknowns];
            cs.weighting_method = wm[j * swm];
            cs.nk = nk[j * snk];
            cs.no = wlsqm_oracle_number_of_dofs(dimension, cs.order);
            if (cs.no < 0 || cs.nk < 0) { bad = 1; no_arr[j] = 0; continue; }
            cs.nr = wlsqm_oracle_number_of_reduced_dofs(cs.no, cs.knowns);
            cs.xi = xi[j * sxi_j];
            cs.yi = (dimension >= 2) ? xi[j * sxi_j + 1] : NAN;   /* infra.pyx:554-556 */
            cs.zi = (dimension == 3) ? xi[j * sxi_j + 2] : NAN;
            size_t need = case_bytes(cs.no, cs.nr > 0 ? cs.nr : 0, cs.nk, do_sens);
            if (need > cap) { free(buf); buf = (char*)malloc(need); cap = need; }
            case_bind(&cs, buf, do_sens);
            int it = fit_one(&cs, xk + j * sxk_j, sxk_k, fk + j * sfk_j, sfk_k, fi + j * sfi_j,
                             sens ? sens + j * ss_j : NULL, ss_k, do_sens && sens, iterative, max_iter, dbg, j);
            if (it > max_it) max_it = it;
            no_arr[j] = cs.no;
            memcpy(fi_stage + j * 35, cs.fi, cs.no * sizeof(double));
        }
        free(buf);
    }
    if (!bad) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(ntasks) schedule(static)
#endif
        for (j = 0; j < ncases; j++)   /* Case_get_fi, infra.pyx:790-795: all `no` entries */
            memcpy(fi + j * sfi_j, fi_stage + j * 35, no_arr[j] * sizeof(double));
    }
    free(fi_stage); free(no_arr);
    return bad ? -2 : max_it;
}

int wlsqm_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
