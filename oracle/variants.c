/* variants.c — attribution study for DESIGN.md section 2 (TEST INFRASTRUCTURE ONLY, like everything under oracle/).
 *
 * One CPU fit routine with switches, so that the distance of each MI355X-first arithmetic choice of the fast kernels from the
 * reference's output can be measured one at a time on the reference-generated goldens (tools/attribution.py):
 *   V_MOMENT  the normal matrix from the distinct moments  mu(P) = sum_k w_k d_k^P  (expanded by 1 / (P_a! P_b!)) instead of
 *             one sum per entry (impl.pyx:566-602)
 *   V_SPLIT   every neighbour sum split over `nsplit` lanes (contiguous slices, met in a butterfly) instead of ascending k
 *   V_FMA     a += b * c contracted to fma() (hipcc default) instead of a rounded product (gcc -O2 x86-64)
 *   V_FASTW   weights from d2 * (1 / max_d2) (one rounded reciprocal) instead of the IEEE quotient (infra.pyx:691-702)
 *   V_LDLT    unscaled, unpivoted LDL^T on the masked full system instead of Ruiz scaling + partial-pivot LU
 *             (lapackdrivers.pyx:553-623, 1628-1665)
 *   V_SYM     upper triangle only, mirrored (the reference sums (w c_m) c_j and (w c_j) c_m separately)
 * With no switch set the routine IS the oracle's arithmetic (checked bit-for-bit against it by the tool).
 * 2D and 3D, any order, knowns mask, WEIGHT_CENTER / UNIFORM. */
#include <math.h>
#include <string.h>

enum { V_MOMENT = 1, V_SPLIT = 2, V_FMA = 4, V_FASTW = 8, V_LDLT = 16, V_SYM = 32 };

static const int P2[15][3] = {{0,0,0},{1,0,0},{0,1,0},{2,0,0},{1,1,0},{0,2,0},{3,0,0},{2,1,0},{1,2,0},{0,3,0},
                              {4,0,0},{3,1,0},{2,2,0},{1,3,0},{0,4,0}};
static const int P3[35][3] = {{0,0,0},{1,0,0},{0,1,0},{0,0,1},{2,0,0},{1,1,0},{0,2,0},{0,1,1},{0,0,2},{1,0,1},
    {3,0,0},{2,1,0},{1,2,0},{0,3,0},{0,2,1},{0,1,2},{0,0,3},{1,0,2},{2,0,1},{1,1,1},
    {4,0,0},{3,1,0},{2,2,0},{1,3,0},{0,4,0},{0,3,1},{0,2,2},{0,1,3},{0,0,4},{1,0,3},{2,0,2},{3,0,1},{2,1,1},{1,2,1},{1,1,2}};
static const double FACT[9] = {1, 1, 2, 6, 24, 120, 720, 5040, 40320};

static double mac(double a, double b, double c, int flags) { return (flags & V_FMA) ? fma(b, c, a) : a + b * c; }

/* sum_k term(k) over nk terms held in t[], ascending or split into nsplit contiguous slices met pairwise */
static double reduce(const double* t, int nk, int flags, int nsplit) {
    if (!(flags & V_SPLIT) || nsplit < 2) { double a = 0.; for (int k = 0; k < nk; k++) a += t[k]; return a; }
    double part[8]; int per = (nk + nsplit - 1) / nsplit;
    for (int s = 0; s < nsplit; s++) { double a = 0.; for (int k = s * per; k < nk && k < (s + 1) * per; k++) a += t[k]; part[s] = a; }
    for (int w = 1; w < nsplit; w *= 2) for (int s = 0; s + w < nsplit; s += 2 * w) part[s] += part[s + w];
    return part[0];
}


static void monomials(int dim, int order, int no, const double* d, double* c) {
    /* same grouping as the oracle's make_c_2D / make_c_3D (impl.pyx:286-432, 70-269) */
    double dx = d[0], dy = d[1], dz = dim == 3 ? d[2] : 0., dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
    const double s6 = 1. / 6., s24 = 1. / 24.;
    c[0] = 1.;
    if (dim == 2) {
        if (order >= 1) { c[1] = dx; c[2] = dy; }
        if (order >= 2) { c[3] = 0.5 * dx2; c[4] = dx * dy; c[5] = 0.5 * dy2; }
        if (order == 3) { c[6] = s6 * dx2 * dx; c[7] = 0.5 * dx2 * dy; c[8] = 0.5 * dx * dy2; c[9] = s6 * dy * dy2; }
        if (order == 4) { double dx3 = dx2 * dx, dy3 = dy2 * dy;
            c[6] = s6 * dx3; c[7] = 0.5 * dx2 * dy; c[8] = 0.5 * dx * dy2; c[9] = s6 * dy3; c[10] = s24 * dx2 * dx2;
            c[11] = s6 * dx3 * dy; c[12] = 0.25 * dx2 * dy2; c[13] = s6 * dx * dy3; c[14] = s24 * dy2 * dy2; }
    } else {
        if (order >= 1) { c[1] = dx; c[2] = dy; c[3] = dz; }
        if (order >= 2) { c[4] = 0.5 * dx2; c[5] = dx * dy; c[6] = 0.5 * dy2; c[7] = dy * dz; c[8] = 0.5 * dz2; c[9] = dx * dz; }
        /* orders 3-4 in 3D are not part of the study */
    }
    (void)no;
}

/* one case; returns 0.  xk [nk, dim], fi in/out [no] */
static void fit_case(int dim, int order, int no, int nk, const double* xk, const double* fk, const double* xi, double* fi,
                     long long knowns, int wm, int flags, int nsplit) {
    const int (*P)[3] = dim == 2 ? P2 : P3;
    double c[128][35], w[128], t[128], M[35][35], g[35];
    double max_d2 = 0.;
    for (int k = 0; k < nk; k++) {
        double d[3] = {0, 0, 0};
        for (int m = 0; m < dim; m++) d[m] = xk[k * dim + m] - xi[m];
        monomials(dim, order, no, d, c[k]);
        double d2 = d[0] * d[0] + d[1] * d[1]; if (dim == 3) d2 += d[2] * d[2];
        w[k] = d2; if (d2 > max_d2) max_d2 = d2;
    }
    const double inv = 1. / max_d2;
    for (int k = 0; k < nk; k++) {
        if (wm == 1) { w[k] = 1.; continue; }
        double q = (flags & V_FASTW) ? w[k] * inv : w[k] / max_d2;
        double tmp = 1. - sqrt(q);
        w[k] = 1e-4 + (1. - 1e-4) * tmp * tmp;
    }
    if (flags & V_MOMENT) {
        /* distinct moments mu(p,q,r) = sum w dx^p dy^q dz^r, then M[a][b] = mu(Pa+Pb) / (Pa! Pb!) */
        double mu[9][9][9];
        int D = 2 * order;
        for (int p = 0; p <= D; p++) for (int q = 0; p + q <= D; q++) for (int r = 0; p + q + r <= (dim == 3 ? D : p + q); r++) {
            for (int k = 0; k < nk; k++) {
                double d[3] = {0, 0, 0};
                for (int m = 0; m < dim; m++) d[m] = xk[k * dim + m] - xi[m];
                double v = w[k];
                for (int e = 0; e < p; e++) v *= d[0];
                for (int e = 0; e < q; e++) v *= d[1];
                for (int e = 0; e < r; e++) v *= d[2];
                t[k] = v;
            }
            mu[p][q][r] = reduce(t, nk, flags, nsplit);
        }
        for (int a = 0; a < no; a++) for (int b = 0; b < no; b++) {
            double den = FACT[P[a][0]] * FACT[P[a][1]] * FACT[P[a][2]] * FACT[P[b][0]] * FACT[P[b][1]] * FACT[P[b][2]];
            M[a][b] = mu[P[a][0] + P[b][0]][P[a][1] + P[b][1]][P[a][2] + P[b][2]] / den;
        }
    } else {
        for (int a = 0; a < no; a++) for (int b = 0; b < no; b++) {
            if ((flags & V_SYM) && b < a) { M[a][b] = M[b][a]; continue; }
            if ((flags & V_SPLIT) || !(flags & V_FMA)) {
                for (int k = 0; k < nk; k++) t[k] = w[k] * c[k][b] * c[k][a];      /* impl.pyx:601: (w c_om) c_oj */
                if (!(flags & V_FMA)) { M[a][b] = reduce(t, nk, flags, nsplit); continue; }
            }
            /* contracted: acc = fma(w c_b, c_a, acc) per slice */
            int ns = (flags & V_SPLIT) ? nsplit : 1, per = (nk + ns - 1) / ns; double part[8];
            for (int s = 0; s < ns; s++) { double acc = 0.; for (int k = s * per; k < nk && k < (s + 1) * per; k++) acc = fma(w[k] * c[k][b], c[k][a], acc); part[s] = acc; }
            for (int ww = 1; ww < ns; ww *= 2) for (int s = 0; s + ww < ns; s += 2 * ww) part[s] += part[s + ww];
            M[a][b] = part[0];
        }
    }
    for (int a = 0; a < no; a++) {
        if (flags & V_FMA) { double acc = 0.; for (int k = 0; k < nk; k++) acc = fma(w[k] * fk[k], c[k][a], acc); g[a] = acc; }
        else { for (int k = 0; k < nk; k++) t[k] = w[k] * fk[k] * c[k][a]; g[a] = reduce(t, nk, flags, nsplit); }
    }
    int r2o[35], nr = 0;
    for (int a = 0; a < no; a++) if (!((knowns >> a) & 1)) r2o[nr++] = a;
    /* infra.pyx:119-121: the reference's nr = no - popcount(knowns) counts stray bits beyond the DOFs too, its remap (:145-200) does not:
     * the FIRST nr unknown DOFs are solved for, the others are never touched (as wlsqm_oracle.c does it) */
    nr = no - __builtin_popcountll((unsigned long long)knowns);
    if (nr < 1) return;
    if (flags & V_LDLT) {
        /* knowns to the right-hand side through the assembled matrix, masked full system, unpivoted LDL^T */
        double A[35][35], b[35];
        for (int j = 0; j < nr; j++) {
            b[j] = g[r2o[j]];
            for (int om = 0; om < no; om++) if ((knowns >> om) & 1) b[j] = mac(b[j], -M[r2o[j]][om], fi[om], flags);
            for (int m = 0; m < nr; m++) A[j][m] = M[r2o[j]][r2o[m]];
        }
        for (int j = 0; j < nr; j++) {
            double invp = 1. / A[j][j];
            for (int i = j + 1; i < nr; i++) {
                double l = A[j][i] * invp;
                for (int m = i; m < nr; m++) A[i][m] = mac(A[i][m], -l, A[j][m], flags);
                A[j][i] = l;
            }
            A[j][j] = invp;
        }
        for (int j = 0; j < nr; j++) for (int i = j + 1; i < nr; i++) b[i] = mac(b[i], -A[j][i], b[j], flags);
        for (int j = nr - 1; j >= 0; j--) { double v = b[j] * A[j][j]; for (int i = j + 1; i < nr; i++) v = mac(v, -A[j][i], b[i], flags); b[j] = v; }
        for (int j = 0; j < nr; j++) fi[r2o[j]] = b[j];
        return;
    }
    /* the reference's own tail: Ruiz + pivoted LU + solve, as in wlsqm_oracle.c */
    double A[35 * 35], rs[35], cs[35], DR[35], DC[35], DRp[35], DCp[35], b[35]; int ipiv[35];
    for (int j = 0; j < nr; j++) for (int m = 0; m < nr; m++) A[j + nr * m] = M[r2o[j]][r2o[m]];
    for (int i = 0; i < nr; i++) rs[i] = cs[i] = DRp[i] = DCp[i] = 1.;
    for (int it = 0; it < 100; it++) {
        for (int j = 0; j < nr; j++) { double acc = 0.; for (int m = 0; m < nr; m++) { double q = fabs(A[j + nr * m] / (DRp[j] * DCp[m])); if (q > acc) acc = q; } DR[j] = sqrt(acc); }
        for (int m = 0; m < nr; m++) { double acc = 0.; for (int j = 0; j < nr; j++) { double q = fabs(A[j + nr * m] / (DCp[m] * DRp[j])); if (q > acc) acc = q; } DC[m] = sqrt(acc); }
        for (int j = 0; j < nr; j++) { DRp[j] *= DR[j]; rs[j] /= DR[j]; }
        for (int m = 0; m < nr; m++) { DCp[m] *= DC[m]; cs[m] /= DC[m]; }
        double acc = 0.; for (int j = 0; j < nr; j++) { double q = fabs(1. - DR[j] * DR[j]); if (q > acc) acc = q; }
        if (acc < 1e-15) { acc = 0.; for (int m = 0; m < nr; m++) { double q = fabs(1. - DC[m] * DC[m]); if (q > acc) acc = q; } if (acc < 1e-15) break; }
    }
    for (int m = 0; m < nr; m++) for (int j = 0; j < nr; j++) A[j + nr * m] *= (rs[j] * cs[m]);
    for (int j = 0; j < nr; j++) {
        int p = j; double best = fabs(A[j + nr * j]);
        for (int i = j + 1; i < nr; i++) if (fabs(A[i + nr * j]) > best) { best = fabs(A[i + nr * j]); p = i; }
        ipiv[j] = p;
        if (A[p + nr * j] != 0.) {
            if (p != j) for (int m = 0; m < nr; m++) { double tt = A[j + nr * m]; A[j + nr * m] = A[p + nr * m]; A[p + nr * m] = tt; }
            double r = 1. / A[j + nr * j]; for (int i = j + 1; i < nr; i++) A[i + nr * j] *= r;
        }
        for (int m = j + 1; m < nr; m++) { double u = A[j + nr * m]; for (int i = j + 1; i < nr; i++) A[i + nr * m] = mac(A[i + nr * m], -A[i + nr * j], u, flags); }
    }
    for (int j = 0; j < nr; j++) b[j] = rs[j] * g[r2o[j]];
    for (int om = 0; om < no; om++) if ((knowns >> om) & 1)
        for (int j = 0; j < nr; j++) for (int k = 0; k < nk; k++) b[j] -= fi[om] * w[k] * c[k][om] * c[k][r2o[j]] * rs[j];
    for (int i = 0; i < nr; i++) if (ipiv[i] != i) { double tt = b[i]; b[i] = b[ipiv[i]]; b[ipiv[i]] = tt; }
    for (int j = 0; j < nr; j++) for (int i = j + 1; i < nr; i++) b[i] = mac(b[i], -A[i + nr * j], b[j], flags);
    for (int j = nr - 1; j >= 0; j--) { b[j] /= A[j + nr * j]; for (int i = 0; i < j; i++) b[i] = mac(b[i], -A[i + nr * j], b[j], flags); }
    for (int j = 0; j < nr; j++) fi[r2o[j]] = b[j] * cs[j];
}

int wlsqm_variant_fit_many(int dim, int order, int no, long ncases, int nk, const double* xk, const double* fk, const double* xi,
                           double* fi, long long knowns, int wm, int flags, int nsplit) {
    if (nk > 128 || no > 35 || (dim != 2 && dim != 3)) return -1;
#pragma omp parallel for schedule(static)
    for (long j = 0; j < ncases; j++)
        fit_case(dim, order, no, nk, xk + j * nk * dim, fk + j * nk, xi + j * dim, fi + j * no, knowns, wm, flags, nsplit);
    return 0;
}

/* Per-case nk / knowns / weighting (rows of max_nk slots, fi rows of fi_stride doubles): the checker of the ACCURATE numerics mode
 * (csrc/fit_accurate.hip = this routine with V_SYM, bit for bit; tests/test_gpu_accurate.py). */
int wlsqm_variant_fit_many_ragged(int dim, int order, int no, long ncases, int max_nk, const double* xk, const double* fk,
                                  const int* nk, const double* xi, double* fi, long fi_stride, const long long* knowns,
                                  const int* wm, int flags, int nsplit) {
    if (max_nk > 128 || no > 35 || (dim != 2 && dim != 3)) return -1;
#pragma omp parallel for schedule(static)
    for (long j = 0; j < ncases; j++) {
        int n = nk[j] < max_nk ? nk[j] : max_nk;
        fit_case(dim, order, no, n, xk + j * (long)max_nk * dim, fk + j * (long)max_nk, xi + j * dim, fi + j * fi_stride, knowns[j], wm[j],
                 flags, nsplit);
    }
    return 0;
}
