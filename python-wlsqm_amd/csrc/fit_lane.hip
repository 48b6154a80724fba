// fit_lane.hip — "one lane per case" WLSQM fit kernel for gfx950 (wave64).
//
// Replaces the loop body of simple.pyx:996-1008 (generic_fit_basic_many_parallel) and
// :1109-1123 (iterative), i.e. make_c_nD + make_A + preprocess_A + solve[_iterative]
// (impl.pyx:47-53, 566-602, 620-689, 731-846, 986-1083) for systems with no <= 15 DOFs.
//
// Mapping: each lane owns one local fit.  The packed upper triangle of the no x no normal
// matrix (<= 120 doubles) plus the right-hand side stay in VGPRs from the first neighbour
// to the back-substitution; nothing but the inputs and the `no` results touches memory.
// This generic variant reads the case's rows straight from global memory with arbitrary
// strides (the reference's memoryview contract); the contiguous fast path lives in
// fit_tile.hip.
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

namespace wlsqm {

constexpr int LANE_BLOCK = 64;   // one wave per workgroup: no cross-wave coupling, max scheduling freedom

// Row access of one case: dense (xk/fk rows with strides) or index-based (hoods row into the S/F point tables).
template <int DIM>
struct CaseRows {
    const double* xr; long long sxk_k;
    const double* fr; long long sfk_k;
    const int* hr; const double* S; const double* F;      // hr != nullptr: index-based
    __device__ __forceinline__ void offset(int k, const double (&xi)[DIM], double (&d)[DIM]) const {
        const double* q = hr ? S + (long long)hr[k] * DIM : xr + k * sxk_k;
#pragma unroll
        for (int m = 0; m < DIM; ++m) d[m] = q[m] - xi[m];
    }
    __device__ __forceinline__ double value(int k) const { return hr ? F[hr[k]] : fr[k * sfk_k]; }
};

template <int DIM, int ORDER, bool EXTRAS>
__global__ __launch_bounds__(LANE_BLOCK) void fit_lane_kernel(const KParams p) {
    constexpr int NO = ndofs(DIM, ORDER);
    constexpr int NE = NO * (NO + 1) / 2;
    const long long t = (long long)blockIdx.x * LANE_BLOCK + threadIdx.x;
    if (t >= live_cases(p)) return;
    const long long j = p.case_index ? p.case_index[t] : t;

    const int nk = min(p.nk[j * p.snk], (int)p.max_nk);      // never past the end of a row
    const bool uniform = (p.wm[j * p.swm] == WLSQM_WEIGHT_UNIFORM);
    unsigned long long known, dropped;
    effective_mask<NO>(p.knowns[j * p.sknowns], known, dropped);
    constexpr unsigned long long FULL = (1ull << NO) - 1ull;
    if (known == FULL) return;                      // nr < 1: no-op (impl.pyx:574, 636, 742)

    double xi[DIM];
    CaseRows<DIM> rows;
    if (p.hoods) {
        const long long pj = own_point(p, j);
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.S[pj * DIM + m];
        rows = CaseRows<DIM>{nullptr, 0, nullptr, 0, p.hoods + j * p.shoods_j, p.S, p.F};
    } else {
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.xi[j * p.sxi_j + m];
        rows = CaseRows<DIM>{p.xk + j * p.sxk_j, p.sxk_k, p.fk + j * p.sfk_j, p.sfk_k, nullptr, nullptr, nullptr};
    }
    double* fio = p.fi + j * p.sfi_j;

    // pass 1: largest squared distance (impl.pyx:389-391 etc.); not needed for uniform weights
    double max_d2 = 0.0;
    if (!uniform) {
        for (int k = 0; k < nk; ++k) {
            double d[DIM];
            rows.offset(k, xi, d);
            double d2 = d[0] * d[0];
            if constexpr (DIM >= 2) d2 += d[1] * d[1];
            if constexpr (DIM == 3) d2 += d[2] * d[2];
            if (d2 > max_d2) max_d2 = d2;
        }
    }
    const double inv_max = inverse_max(max_d2);

    // pass 2: M = C^T W C (upper triangle), g = C^T W f
    double M[NE], g[NO];
#pragma unroll
    for (int e = 0; e < NE; ++e) M[e] = 0.0;
#pragma unroll
    for (int a = 0; a < NO; ++a) g[a] = 0.0;
    for (int k = 0; k < nk; ++k) {
        double d[DIM], c[NO];
        rows.offset(k, xi, d);
        const double d2 = monomials<DIM, ORDER>(d, c);
        const double w = weight(d2, inv_max, uniform);
        accumulate<NO>(M, g, c, w, rows.value(k));
    }

    // knowns: values come from the user's fi (Case_set_fi, infra.pyx:780-785)
    double val[NO];
#pragma unroll
    for (int a = 0; a < NO; ++a) val[a] = ((known & ~dropped) >> a) & 1ull ? fio[a] : 0.0;
    eliminate_knowns<NO>(M, g, known, val);

    ldlt_factor<NO>(M);
    ldlt_solve<NO>(M, g);

    if constexpr (!EXTRAS) {
#pragma unroll
        for (int a = 0; a < NO; ++a)
            if (!((known >> a) & 1ull)) fio[a] = g[a];
        return;
    } else {
        // ---- sensitivities (impl.pyx:776-778, 821-823, 831-846): sens[k,a] = d fi[a] / d fk[k]
        if (p.do_sens && p.sens) {
            double* sr = p.sens + j * p.ss_j;
            for (int k = 0; k < nk; ++k) {
                double d[DIM], c[NO], s[NO];
                rows.offset(k, xi, d);
                const double d2 = monomials<DIM, ORDER>(d, c);
                const double w = weight(d2, inv_max, uniform);
#pragma unroll
                for (int a = 0; a < NO; ++a) s[a] = ((known >> a) & 1ull) ? 0.0 : ((a == 0) ? w : w * c[a]);
                ldlt_solve<NO>(M, s);
#pragma unroll
                for (int a = 0; a < NO; ++a) {
                    if (!((known >> a) & 1ull)) sr[k * p.ss_k + a] = s[a];
                    else if (!((dropped >> a) & 1ull)) sr[k * p.ss_k + a] = __longlong_as_double(0x7ff8000000000000LL);
                }
            }
        }
        // ---- iterative refinement (impl.pyx:986-1083)
        int iters = 0;
        if (p.iterative) {
            double fi[NO];
#pragma unroll
            for (int a = 0; a < NO; ++a) fi[a] = ((known >> a) & 1ull) ? val[a] : g[a];
            // dropped DOFs keep whatever the user's fi holds for the model evaluation (Case_set_fi copies all `no`)
#pragma unroll
            for (int a = 0; a < NO; ++a) if ((dropped >> a) & 1ull) fi[a] = fio[a];
            double prev_norm = -1.0;
            bool broke = false;
            int i = 0;
            for (i = 0; i < p.max_iter; ++i) {
                double norm = 0.0;
                double r[NO];
#pragma unroll
                for (int a = 0; a < NO; ++a) r[a] = 0.0;
                for (int k = 0; k < nk; ++k) {
                    double d[DIM], c[NO];
                    rows.offset(k, xi, d);
                    const double d2 = monomials<DIM, ORDER>(d, c);
                    const double w = weight(d2, inv_max, uniform);
                    double model = fi[0];                       // taylor_*D (polyeval.pyx): sum_a c[a] fi[a]
#pragma unroll
                    for (int a = 1; a < NO; ++a) model += c[a] * fi[a];
                    const double res = rows.value(k) - model;
                    const double ar = fabs(res);
                    if (k == 0) norm = ar; else if (ar > norm) norm = ar;   // impl.pyx:1037-1041
                    const double wr = w * res;
#pragma unroll
                    for (int a = 0; a < NO; ++a) r[a] += (a == 0) ? wr : wr * c[a];
                }
                if (norm == prev_norm) { broke = true; break; }  // impl.pyx:1057
                prev_norm = norm;
#pragma unroll
                for (int a = 0; a < NO; ++a) if ((known >> a) & 1ull) r[a] = 0.0;   // knowns of the correction are 0
                ldlt_solve<NO>(M, r);
#pragma unroll
                for (int a = 0; a < NO; ++a) if (!((known >> a) & 1ull)) fi[a] += r[a];
            }
            iters = broke ? i : (p.max_iter > 0 ? p.max_iter : 1);   // for/else, impl.pyx:1080-1081
#pragma unroll
            for (int a = 0; a < NO; ++a) g[a] = fi[a];
        }
#pragma unroll
        for (int a = 0; a < NO; ++a)
            if (!((known >> a) & 1ull)) fio[a] = g[a];
        if (p.iterative && p.iters_out) atomicMax(p.iters_out, iters);
    }
}

template <int DIM, int ORDER>
static int launch_lane(const KParams& p, hipStream_t stream) {
    const long long blocks = (p.ncases + LANE_BLOCK - 1) / LANE_BLOCK;
    if (blocks <= 0) return WLSQM_OK;
    if (blocks > 0x7fffffffLL) { set_error("too many cases for one launch"); return WLSQM_EVALUE; }
    const bool extras = p.do_sens || p.iterative;
    if (extras)
        hipLaunchKernelGGL((fit_lane_kernel<DIM, ORDER, true>), dim3((unsigned)blocks), dim3(LANE_BLOCK), 0, stream, p);
    else
        hipLaunchKernelGGL((fit_lane_kernel<DIM, ORDER, false>), dim3((unsigned)blocks), dim3(LANE_BLOCK), 0, stream, p);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("lane");
    return WLSQM_OK;
}

int launch_fit_lane(int dimension, int order, const KParams& p, hipStream_t stream) {
#define CASE(D, O) if (dimension == D && order == O) return launch_lane<D, O>(p, stream);
    CASE(1, 0) CASE(1, 1) CASE(1, 2) CASE(1, 3) CASE(1, 4)
    CASE(2, 0) CASE(2, 1) CASE(2, 2) CASE(2, 3) CASE(2, 4)
    CASE(3, 0) CASE(3, 1) CASE(3, 2)
#undef CASE
    set_error("fit_lane: unsupported (dimension, order)");
    return WLSQM_EVALUE;
}

}  // namespace wlsqm
