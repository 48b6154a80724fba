// fit_wave.hip — "one wavefront per case" WLSQM fit kernel for the large systems
// (3D order 3: no = 20, 3D order 4: no = 35; also usable for any other (dim, order)).
//
// Same arithmetic as fit_lane.hip (see wlsqm_kernels.hpp for the reference citations), but
// the no x no normal matrix does not fit one lane's registers, so the 64 lanes of a wave
// cooperate through LDS:
//   * neighbours are processed in chunks of 64: lane k builds the monomial row c[k,:] and
//     the weight of its neighbour and parks them in LDS;
//   * the no(no+1)/2 unique entries of M = C^T W C are dealt round-robin to the lanes, each
//     lane accumulating its <= 10 entries over the chunk (k ascending, as impl.pyx:599-601);
//   * knowns are masked to identity rows/columns; a left-looking LDL^T runs in LDS with one
//     matrix row per lane (row stride no+2 doubles: conflict-free ds_read_b64);
//   * the triangular solves run redundantly in every lane's registers with broadcast LDS
//     reads of the factor — which also gives 64 right-hand sides at once for the
//     sensitivity pass (impl.pyx:831-834 loops dgetrs over the nk right-hand sides).
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

namespace wlsqm {

constexpr int WAVE = 64;

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(v, off, WAVE);
        v = (o > v) ? o : v;
    }
    return v;
}

// Cooperative single-RHS solve with the LDS-resident factor (unit lower L strictly below the
// diagonal, 1/d on it): lane i owns b[i]; one column per step.  Replaces dgetrs
// (lapackdrivers.pyx:1657-1665).  Must be called by the whole workgroup.
template <int N, int LD>
__device__ __forceinline__ void coop_ldlt_solve(const double* L, double* b, int lane) {
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
        __syncthreads();
        if (lane > j && lane < N) b[lane] -= L[lane * LD + j] * b[j];
    }
    __syncthreads();
    if (lane < N) b[lane] *= L[lane * LD + lane];
#pragma unroll 1
    for (int j = N - 1; j >= 0; --j) {
        __syncthreads();
        if (lane < j) b[lane] -= L[j * LD + lane] * b[j];
    }
    __syncthreads();
}

// 64 right-hand sides at once: lane l owns column l of X[N][WAVE] (LDS); the factor is read with
// broadcast ds_reads.  No cross-lane dependency, so no barriers.
template <int N, int LD>
__device__ __forceinline__ void column_ldlt_solve(const double* L, double* X, int lane) {
    // rolled on purpose: fully unrolled, hipcc hoists hundreds of ds_reads and spills to scratch
#pragma unroll 1
    for (int j = 0; j < N; ++j) {
        const double bj = X[j * WAVE + lane];
#pragma unroll 4
        for (int i = j + 1; i < N; ++i) X[i * WAVE + lane] -= L[i * LD + j] * bj;
    }
#pragma unroll 1
    for (int j = N - 1; j >= 0; --j) {
        double v = X[j * WAVE + lane] * L[j * LD + j];
#pragma unroll 4
        for (int i = j + 1; i < N; ++i) v -= L[i * LD + j] * X[i * WAVE + lane];
        X[j * WAVE + lane] = v;
    }
}

template <int DIM, int ORDER>
__global__ __launch_bounds__(WAVE) void fit_wave_kernel(const KParams p) {
    constexpr int NO = ndofs(DIM, ORDER);
    constexpr int NE = NO * (NO + 1) / 2;
    constexpr int EPL = (NE + WAVE - 1) / WAVE;      // entries per lane
    constexpr int LD = NO + 2;                       // row stride of the LDS matrix (doubles)
    constexpr int LC = NO + 1;                       // row stride of the chunk's c rows
    __shared__ double sC[WAVE * LC];
    __shared__ double sW[WAVE];
    __shared__ double sWF[WAVE];
    __shared__ double sM[NO * LD];
    __shared__ double sB[NO];
    __shared__ double sU[NO];
    __shared__ double sVal[NO];

    const int lane = threadIdx.x;
    const long long t = blockIdx.x;
    if (t >= live_cases(p)) return;                               // (block-uniform: before any barrier)
    const long long j = p.case_index ? p.case_index[t] : t;
    const int nk = min(p.nk[j * p.snk], (int)p.max_nk);      // never past the end of a row
    const bool uniform = (p.wm[j * p.swm] == WLSQM_WEIGHT_UNIFORM);
    unsigned long long known, dropped;
    effective_mask<NO>(p.knowns[j * p.sknowns], known, dropped);
    constexpr unsigned long long FULL = (NO >= 64) ? ~0ull : ((1ull << NO) - 1ull);
    if (known == FULL) return;                        // whole block exits together

    double xi[DIM];
#pragma unroll
    for (int m = 0; m < DIM; ++m) xi[m] = p.xi[j * p.sxi_j + m];
    const double* xr = p.xk + j * p.sxk_j;
    const double* fr = p.fk + j * p.sfk_j;
    double* fio = p.fi + j * p.sfi_j;

    // pass 1: max squared distance
    double max_d2 = 0.0;
    if (!uniform) {
        for (int k = lane; k < nk; k += WAVE) {
            double d2 = 0.0;
#pragma unroll
            for (int m = 0; m < DIM; ++m) { const double dd = xr[k * p.sxk_k + m] - xi[m]; d2 += dd * dd; }
            if (d2 > max_d2) max_d2 = d2;
        }
        max_d2 = wave_max(max_d2);
    }
    const double inv_max = inverse_max(max_d2);

    // this lane's entries of the packed upper triangle
    int ea[EPL], eb[EPL];
    double acc[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        int e = lane + WAVE * i;
        if (e >= NE) e = NE - 1;                      // clamp: harmless duplicate, never stored
        int a = 0, rem = e;
        while (rem >= NO - a) { rem -= NO - a; ++a; }
        ea[i] = a; eb[i] = a + rem; acc[i] = 0.0;
    }
    double gacc = 0.0;

    // pass 2: chunks of 64 neighbours
    for (int kb = 0; kb < nk; kb += WAVE) {
        const int kc = min(WAVE, nk - kb);
        if (lane < kc) {
            double d[DIM], c[NO];
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = xr[(kb + lane) * p.sxk_k + m] - xi[m];
            const double d2 = monomials<DIM, ORDER>(d, c);
            const double w = weight(d2, inv_max, uniform);
#pragma unroll
            for (int a = 0; a < NO; ++a) sC[lane * LC + a] = c[a];
            sW[lane] = w;
            sWF[lane] = w * fr[(kb + lane) * p.sfk_k];
        }
        __syncthreads();
        for (int k = 0; k < kc; ++k) {
            const double w = sW[k];
            const double* ck = &sC[k * LC];
#pragma unroll
            for (int i = 0; i < EPL; ++i) acc[i] += (w * ck[ea[i]]) * ck[eb[i]];
            if (lane < NO) gacc += sWF[k] * ck[lane];
        }
        __syncthreads();
    }

    // scatter to the full symmetric LDS matrix
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        if (lane + WAVE * i < NE) {
            sM[ea[i] * LD + eb[i]] = acc[i];
            sM[eb[i] * LD + ea[i]] = acc[i];
        }
    }
    const bool mine = lane < NO;
    const bool kn_me = mine && ((known >> lane) & 1ull);
    if (mine) {
        sB[lane] = gacc;
        sVal[lane] = (((known & ~dropped) >> lane) & 1ull) ? fio[lane] : 0.0;
    }
    __syncthreads();
    // knowns elimination (impl.pyx:792-818), ascending om
    if (mine && !kn_me) {
        double bb = sB[lane];
        for (int om = 0; om < NO; ++om)
            if ((known >> om) & 1ull) bb -= sM[lane * LD + om] * sVal[om];
        sB[lane] = bb;
    }
    __syncthreads();
    if (mine) {
        for (int m = 0; m < NO; ++m)
            if (kn_me || ((known >> m) & 1ull)) sM[lane * LD + m] = (m == lane) ? 1.0 : 0.0;
        if (kn_me) sB[lane] = 0.0;
    }
    __syncthreads();

    // left-looking LDL^T, one row per lane; afterwards: unit L strictly below the diagonal, 1/d on it
    for (int c = 0; c < NO; ++c) {
        if (lane < c) sU[lane] = sM[c * LD + lane] / sM[lane * LD + lane];     // L[c][m] * d_m  (diag holds 1/d_m)
        __syncthreads();
        double v = 0.0;
        if (mine && lane >= c) {
            v = sM[lane * LD + c];
            for (int m = 0; m < c; ++m) v -= sM[lane * LD + m] * sU[m];
        }
        if (lane == c) sM[c * LD + c] = 1.0 / v;
        __syncthreads();
        if (mine && lane > c) sM[lane * LD + c] = v * sM[c * LD + c];
        __syncthreads();
    }

    // solve for the right-hand side; the solution stays in LDS (sB), one entry per lane
    coop_ldlt_solve<NO, LD>(sM, sB, lane);

    // ---- inverse of the eliminated normal matrix for fit_sens.hip (p.ws[t][no][no]): lane a solves for unit vector a
    // (zero row and column for a known DOF); the rows leave as they are, the matrix is symmetric
    if (p.ws) {
        double* X = sC;
        __syncthreads();
#pragma unroll 1
        for (int a = 0; a < NO; ++a) X[a * WAVE + lane] = (a == lane && !((known >> a) & 1ull)) ? 1.0 : 0.0;
        column_ldlt_solve<NO, LD>(sM, X, lane);
        if (mine) {
            double* wi = p.ws + (t * NO + lane) * (long long)NO;
            for (int a = 0; a < NO; ++a) wi[a] = X[a * WAVE + lane];
        }
        __syncthreads();
    } else
    // ---- sensitivities: lane k solves for neighbour k's right-hand side (column k of sC, reused as X[NO][64])
    if (p.do_sens && p.sens) {
        double* sr = p.sens + j * p.ss_j;
        double* X = sC;
        static_assert(NO * WAVE <= WAVE * LC, "sC too small to hold X[NO][64]");
        for (int kb = 0; kb < nk; kb += WAVE) {
            const int k = kb + lane;
            __syncthreads();
            if (k < nk) {
                double d[DIM], c[NO];
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = xr[k * p.sxk_k + m] - xi[m];
                const double d2 = monomials<DIM, ORDER>(d, c);
                const double w = weight(d2, inv_max, uniform);
#pragma unroll
                for (int a = 0; a < NO; ++a) X[a * WAVE + lane] = ((known >> a) & 1ull) ? 0.0 : w * c[a];
                column_ldlt_solve<NO, LD>(sM, X, lane);
                for (int a = 0; a < NO; ++a) {
                    if (!((known >> a) & 1ull)) sr[k * p.ss_k + a] = X[a * WAVE + lane];
                    else if (!((dropped >> a) & 1ull)) sr[k * p.ss_k + a] = __longlong_as_double(0x7ff8000000000000LL);
                }
            }
        }
        __syncthreads();
    }

    // ---- iterative refinement (impl.pyx:986-1083).  sVal holds the full coefficient vector fi.
    int iters = 0;
    if (p.iterative) {
        if (mine) {
            if ((dropped >> lane) & 1ull) sVal[lane] = fio[lane];          // Case_set_fi copies all `no` entries
            else if (!kn_me) sVal[lane] = sB[lane];                          // knowns already hold the user's value
        }
        __syncthreads();
        double prev_norm = -1.0;
        bool broke = false;
        int i = 0;
        for (i = 0; i < p.max_iter; ++i) {
            double norm = 0.0;
            bool first = true;
            double racc = 0.0;                                              // lane a accumulates r[a]
            for (int kb = 0; kb < nk; kb += WAVE) {
                const int kc = min(WAVE, nk - kb);
                double ar = -1.0;                                           // lanes without a neighbour never win the max
                if (lane < kc) {
                    double d[DIM], c[NO];
#pragma unroll
                    for (int m = 0; m < DIM; ++m) d[m] = xr[(kb + lane) * p.sxk_k + m] - xi[m];
                    const double d2 = monomials<DIM, ORDER>(d, c);
                    const double w = weight(d2, inv_max, uniform);
                    double model = sVal[0];                                 // taylor_*D (polyeval.pyx): sum_a c[a] fi[a]
#pragma unroll
                    for (int a = 1; a < NO; ++a) model += c[a] * sVal[a];
                    const double res = fr[(kb + lane) * p.sfk_k] - model;
                    ar = fabs(res);
                    const double wr = w * res;
#pragma unroll
                    for (int a = 0; a < NO; ++a) sC[lane * LC + a] = wr * c[a];
                }
                // max |res| with the NaN semantics of impl.pyx:1037-1041: a NaN in the FIRST residual
                // poisons the norm, later NaNs are ignored.
                const double ar0 = __shfl(ar, 0, WAVE);
                const double cm = wave_max((ar == ar) ? ar : -1.0);
                if (first) { norm = (ar0 == ar0) ? cm : ar0; first = false; }
                else if (cm > norm) norm = cm;
                __syncthreads();
                if (mine)
                    for (int k = 0; k < kc; ++k) racc += sC[k * LC + lane];
                __syncthreads();
            }
            if (norm == prev_norm) { broke = true; break; }                 // impl.pyx:1057
            prev_norm = norm;
            if (mine) sB[lane] = kn_me ? 0.0 : racc;                        // knowns of the correction are 0
            coop_ldlt_solve<NO, LD>(sM, sB, lane);
            if (mine && !kn_me) sVal[lane] += sB[lane];
            __syncthreads();
        }
        iters = broke ? i : (p.max_iter > 0 ? p.max_iter : 1);              // for/else, impl.pyx:1080-1081
        if (mine && !kn_me) sB[lane] = sVal[lane];
        __syncthreads();
    }

    if (mine && !kn_me) fio[lane] = sB[lane];
    if (lane == 0 && p.iterative && p.iters_out) atomicMax(p.iters_out, iters);
}

template <int DIM, int ORDER>
static int launch_wave(const KParams& p, hipStream_t stream) {
    if (p.ncases > 0x7fffffffLL) { set_error("too many cases for one launch"); return WLSQM_EVALUE; }
    hipLaunchKernelGGL((fit_wave_kernel<DIM, ORDER>), dim3((unsigned)p.ncases), dim3(WAVE), 0, stream, p);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("wave");
    return WLSQM_OK;
}

int launch_fit_wave(int dimension, int order, const KParams& p_in, hipStream_t stream) {
    KParams p = p_in;
    p.ws = nullptr;
    if (dimension == 3 && order == 3) return launch_wave<3, 3>(p, stream);
    if (dimension == 3 && order == 4) return launch_wave<3, 4>(p, stream);
    set_error("fit_wave: unsupported (dimension, order)");
    return WLSQM_EVALUE;
}

// The basic fit, which also leaves every case's inverse normal matrix at inv[t][no][no] (first kernel of fit_sens.hip).
int launch_fit_wave_inverse(int dimension, int order, const KParams& p_in, double* inv, hipStream_t stream) {
    KParams p = p_in;
    p.ws = inv; p.do_sens = 0; p.sens = nullptr; p.iterative = 0;
    int rc = WLSQM_EVALUE;
    if (dimension == 3 && order == 3) rc = launch_wave<3, 3>(p, stream);
    else if (dimension == 3 && order == 4) rc = launch_wave<3, 4>(p, stream);
    else set_error("fit_wave_inverse: unsupported (dimension, order)");
    if (rc == WLSQM_OK) note_kernel("wave-inverse");
    return rc;
}

}  // namespace wlsqm
