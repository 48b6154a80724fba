// conds.hip — debug-mode 2-norm condition numbers of the scaled problem matrices,
// ExpertSolver.conds() (expert.pyx:429-464; computed in impl.pyx:662-682 by dgesvd on the
// Ruiz-scaled reduced matrix).
//
// This is a diagnostics path, not a throughput path: one lane per case, the reduced matrix and
// the scaling vectors live in a global workspace laid out [entry][case] (so the lanes of a wave
// touch consecutive addresses), all loops rolled.  Steps, per case:
//   1. reduced normal matrix A[j,m] = sum_k w_k c[k,r2o[m]] c[k,r2o[j]]   (impl.pyx:566-602, via infra.remap)
//   2. Ruiz equilibration exactly as lapackdrivers.pyx:553-623 (eps 1e-15, <= 100 sweeps), applied as :293-299
//   3. singular values by one-sided Jacobi (Hestenes) rotations on the columns; cond = s_max / s_min.
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

namespace wlsqm {

template <int DIM, int ORDER>
__global__ __launch_bounds__(64) void cond_kernel(const KParams p, const int* __restrict__ order_arr,
                                                  double* __restrict__ ws, const long long CH,
                                                  const long long case0, double* __restrict__ out) {
    constexpr int NO = ndofs(DIM, ORDER);
    const long long t = (long long)blockIdx.x * 64 + threadIdx.x;
    if (t >= CH || case0 + t >= p.ncases) return;
    const long long j = case0 + t;
    if (order_arr && order_arr[j] != ORDER) return;                    // heterogeneous batch: another launch owns this case
#define WS(i) ws[(size_t)(i) * CH + t]
    constexpr int oA = 0, oC = NO * NO, oR = oC + NO, oDR = oR + NO, oDC = oDR + NO, oDRp = oDC + NO, oDCp = oDRp + NO,
                  oRS = oDCp + NO, oCS = oRS + NO;

    const int nk = min(p.nk[j * p.snk], (int)p.max_nk);      // never past the end of a row
    const bool uniform = (p.wm[j * p.swm] == WLSQM_WEIGHT_UNIFORM);
    unsigned long long known, dropped;
    effective_mask<NO>(p.knowns[j * p.sknowns], known, dropped);
    int nr = 0;
    for (int a = 0; a < NO; ++a)
        if (!((known >> a) & 1ull)) { WS(oR + nr) = (double)a; ++nr; }     // r2o (infra.pyx:189-192)
    if (nr < 1) { out[j] = __longlong_as_double(0x7ff8000000000000LL); return; }   // cond stays NaN (infra.pyx:577-578)

    double xi[DIM];
#pragma unroll
    for (int m = 0; m < DIM; ++m) xi[m] = p.xi[j * p.sxi_j + m];
    const double* xr = p.xk + j * p.sxk_j;

    double max_d2 = 0.0;
    for (int k = 0; k < nk; ++k) {
        double d2 = 0.0;
#pragma unroll
        for (int m = 0; m < DIM; ++m) { const double dd = xr[k * p.sxk_k + m] - xi[m]; d2 += dd * dd; }
        if (d2 > max_d2) max_d2 = d2;
    }
    const double inv_max = inverse_max(max_d2);

    for (int e = 0; e < nr * nr; ++e) WS(oA + e) = 0.0;
    for (int k = 0; k < nk; ++k) {
        double d[DIM], c[NO];
#pragma unroll
        for (int m = 0; m < DIM; ++m) d[m] = xr[k * p.sxk_k + m] - xi[m];
        const double d2 = monomials<DIM, ORDER>(d, c);
        const double w = weight(d2, inv_max, uniform);
#pragma unroll
        for (int a = 0; a < NO; ++a) WS(oC + a) = c[a];
        for (int mm = 0; mm < nr; ++mm) {
            const double wc = w * WS(oC + (int)WS(oR + mm));
            for (int jj = 0; jj < nr; ++jj) WS(oA + jj + nr * mm) += wc * WS(oC + (int)WS(oR + jj));
        }
    }

    // Ruiz (2001) equilibration, lapackdrivers.pyx:553-623
    for (int i = 0; i < nr; ++i) { WS(oRS + i) = 1.0; WS(oCS + i) = 1.0; WS(oDRp + i) = 1.0; WS(oDCp + i) = 1.0; }
    for (int it = 0; it < 100; ++it) {
        for (int jj = 0; jj < nr; ++jj) {
            const double r = WS(oDRp + jj);
            double acc = 0.0;
            for (int mm = 0; mm < nr; ++mm) { const double v = fabs(WS(oA + jj + nr * mm) / (r * WS(oDCp + mm))); if (v > acc) acc = v; }
            WS(oDR + jj) = sqrt(acc);
        }
        for (int mm = 0; mm < nr; ++mm) {
            const double cc = WS(oDCp + mm);
            double acc = 0.0;
            for (int jj = 0; jj < nr; ++jj) { const double v = fabs(WS(oA + jj + nr * mm) / (cc * WS(oDRp + jj))); if (v > acc) acc = v; }
            WS(oDC + mm) = sqrt(acc);
        }
        double er = 0.0, ec = 0.0;
        for (int i = 0; i < nr; ++i) {
            const double dr = WS(oDR + i), dc = WS(oDC + i);
            WS(oDRp + i) *= dr; WS(oRS + i) /= dr;
            WS(oDCp + i) *= dc; WS(oCS + i) /= dc;
            const double a1 = fabs(1.0 - dr * dr), a2 = fabs(1.0 - dc * dc);
            if (a1 > er) er = a1;
            if (a2 > ec) ec = a2;
        }
        if (er < 1e-15 && ec < 1e-15) break;
    }
    for (int mm = 0; mm < nr; ++mm) {                                   // apply_scaling_c, lapackdrivers.pyx:293-299
        const double cc = WS(oCS + mm);
        for (int jj = 0; jj < nr; ++jj) WS(oA + jj + nr * mm) *= (WS(oRS + jj) * cc);
    }

    // one-sided Jacobi SVD: orthogonalise the columns; singular values = column norms
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int pp = 0; pp < nr - 1; ++pp) {
            for (int q = pp + 1; q < nr; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int i = 0; i < nr; ++i) {
                    const double ap = WS(oA + i + nr * pp), aq = WS(oA + i + nr * q);
                    alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
                }
                if (fabs(gamma) > 1e-15 * sqrt(alpha * beta) && gamma != 0.0) {
                    const double zeta = (beta - alpha) / (2.0 * gamma);
                    const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                    for (int i = 0; i < nr; ++i) {
                        const double ap = WS(oA + i + nr * pp), aq = WS(oA + i + nr * q);
                        WS(oA + i + nr * pp) = cs * ap - sn * aq;
                        WS(oA + i + nr * q) = sn * ap + cs * aq;
                    }
                    rotated = true;
                }
            }
        }
        if (!rotated) break;
    }
    double smax = 0.0, smin = 1.0 / 0.0;
    for (int q = 0; q < nr; ++q) {
        double s = 0.0;
        for (int i = 0; i < nr; ++i) { const double a = WS(oA + i + nr * q); s += a * a; }
        s = sqrt(s);
        if (s > smax) smax = s;
        if (s < smin) smin = s;
    }
    out[j] = smax / smin;                                               // impl.pyx:680
#undef WS
}

template <int DIM, int ORDER>
static int launch_cond(const KParams& p, const int* order_arr, double* ws, long long CH, long long case0, double* out,
                       hipStream_t stream) {
    const long long blocks = (CH + 63) / 64;
    hipLaunchKernelGGL((cond_kernel<DIM, ORDER>), dim3((unsigned)blocks), dim3(64), 0, stream, p, order_arr, ws, CH, case0, out);
    WLSQM_HIP_CHECK(hipGetLastError());
    return WLSQM_OK;
}

// workspace doubles per case for (dim, order)
long long cond_workspace_doubles(int no) { return (long long)no * no + 8LL * no; }

// Condition numbers of the cases in [case0, case0 + CH) whose order is `order` (order_arr == nullptr: all of them).
int launch_conds(int dimension, int order, const KParams& p, const int* order_arr, double* ws, long long CH,
                 long long case0, double* out, hipStream_t stream) {
#define CASE(D, O) if (dimension == D && order == O) return launch_cond<D, O>(p, order_arr, ws, CH, case0, out, stream);
    CASE(1, 0) CASE(1, 1) CASE(1, 2) CASE(1, 3) CASE(1, 4)
    CASE(2, 0) CASE(2, 1) CASE(2, 2) CASE(2, 3) CASE(2, 4)
    CASE(3, 0) CASE(3, 1) CASE(3, 2) CASE(3, 3) CASE(3, 4)
#undef CASE
    set_error("conds: unsupported (dimension, order)");
    return WLSQM_EVALUE;
}

}  // namespace wlsqm
