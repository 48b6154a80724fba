// fit_chunk.hip — basic fit for ANY number of neighbour slots (K > 128, and the (dimension, order, K) combinations without a
// fixed-K or runtime-K tile kernel): the neighbours of a 16-case tile travel through LDS in chunks of 32 slots, twice — once for
// the largest squared distance of every case (the weights need it, infra.pyx:668-702), once for the moments.  1.6x the
// algorithmic traffic on xk, against the lane-per-case kernel's uncoalesced row reads (8.5 % of the HBM peak on C2).
//
// Same arithmetic as the tile kernels (wlsqm_tile.hpp): one wave per tile, four lanes per case (lane = h * 16 + c), every lane
// takes 8 of a chunk's 32 slots, moments (wlsqm_moments.hpp), xor butterfly, lane h == 0 expands, eliminates knowns
// (impl.pyx:792-823), factors and substitutes.  Dense contiguous rows with an even K (api.hip repacks anything else).
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_moments.hpp"

namespace wlsqm {

typedef double cd2_ __attribute__((ext_vector_type(2)));

// INV: besides fi, every case leaves the inverse of its (knowns-eliminated) normal matrix at p.ws[j][no][no] for the
// sensitivities (fit_sens.hip): after the butterfly the four lanes of a case hold the same sums, so each factors the matrix
// for itself and substitutes its share of the unit vectors (columns h, h + 4, ...).
// ITER (round 3): the iterative refinement of solve_iterative (impl.pyx:986-1083) on the same tile — every lane of a case keeps
// the LDL^T factor of its case in registers (as for INV), and every sweep streams the tile's chunks through LDS once more: each
// lane evaluates the model at its 8 slots of a chunk, takes the residual and its share of C^T W res; the four lanes meet in the
// butterfly (maximum norm, correction's right-hand side), each substitutes for itself.  The stop test is the reference's: exact
// equality of two successive residual norms, per case; the wave sweeps until its last case has stopped.  Replaces the
// lane-per-case kernel (uncoalesced rows, 3.7 % of the HBM peak on configs[2] with refinement) for the 10- and 15-unknown systems.
template <int DIM, int ORDER, int MINW, bool INV = false, bool ITER = false>
__global__ __launch_bounds__(64, MINW) void fit_chunk_kernel(const KParams p, const long long ntiles, const int K, const bool cache_x) {
    constexpr int WV = 64, TC = 16, LPC = 4, CH = 32, SPL = CH / LPC;          // slots per lane and chunk
    constexpr int NO = ndofs(DIM, ORDER), NE = NO * (NO + 1) / 2, NM = mom_count<DIM>(2 * ORDER);
    constexpr int CPR = CH * DIM / 2;                                           // 16-byte pieces per row of a chunk
    constexpr int RS = (DIM == 2) ? CH * DIM + 2 : CH * DIM + 1;                // padded row (conflict-free ds_read_b128 / b64)
    constexpr int NX = (TC * CPR + WV - 1) / WV;
    __shared__ __attribute__((aligned(16))) double sX[TC * RS + 2];
    // ITER: the whole tile — xk rows [case][K * DIM] (odd pitch) and fk rows [case][K] (odd pitch) — parked in dynamic LDS by the
    // moment pass and read by every refinement sweep: a sweep then has no global load and no barrier at all (with the chunks
    // re-staged per sweep a lone wave per SIMD paid two exposed load latencies per sweep: 0.19 ms per sweep and 400k cases of
    // 2D order 4, as slow as the lane kernel's uncoalesced rows).  Launched with the cache when four waves per CU still fit
    // (cache_x), without it otherwise (the sweeps re-stage).  (Caching the weights as well measured no gain.)
    extern __shared__ __attribute__((aligned(16))) double sC[];
    const int RSA = (K * DIM) | 1, RFA = K | 1;
    double* const sXall = sC;
    double* const sFall = sC + TC * RSA;
    const int lane = threadIdx.x, c = lane % TC, h = lane / TC;
    const int nchunks = (K + CH - 1) / CH;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long j0 = tile * TC, j = j0 + c;
        const bool valid = j < p.ncases;
        const long long jc = valid ? j : p.ncases - 1;
        const long long nvalid = (p.ncases - j0 < TC) ? (p.ncases - j0) : TC;
        const int nkc = min(p.nk[jc * p.snk], K);
        const bool uniform = (p.wm[jc * p.swm] == WLSQM_WEIGHT_UNIFORM);
        unsigned long long known, dropped;
        effective_mask<NO>(p.knowns[jc * p.sknowns], known, dropped);
        double xi[DIM];
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.xi[jc * p.sxi_j + m];

        // one chunk of the tile's rows into LDS: rows j0 .. j0 + nvalid - 1, slots [k0c, k0c + CH) clipped to the row
        auto stage = [&](int k0c, bool fill = false) {
            const int live_pieces = (min(CH, K - k0c) * DIM) / 2;             // (K * DIM even, k0c * DIM even)
            cd2_ b[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int q = lane + i * WV;
                int r = q / CPR, c2 = q - r * CPR;
                r = r < (int)nvalid ? r : (int)nvalid - 1;
                c2 = c2 < live_pieces ? c2 : live_pieces - 1;
                b[i] = *reinterpret_cast<const cd2_*>(p.xk + ((j0 + r) * (long long)K + k0c) * DIM + 2 * c2);
            }
            __syncthreads();                                                   // the previous chunk has been consumed
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int q = lane + i * WV;
                if (TC * CPR % WV == 0 || q < TC * CPR) {
                    const int r = q / CPR, c2 = q - r * CPR;
                    double* d = sX + r * RS + 2 * c2;
                    if constexpr (RS % 2 == 0) *reinterpret_cast<cd2_*>(d) = b[i];
                    else { d[0] = b[i].x; d[1] = b[i].y; }
                    if constexpr (ITER) {
                        if (fill && 2 * c2 < (K - k0c) * DIM) { double* e = sXall + r * RSA + k0c * DIM + 2 * c2; e[0] = b[i].x; e[1] = b[i].y; }
                    }
                }
            }
            __syncthreads();
        };
        const double* xr = sX + c * RS + h * SPL * DIM;                        // this lane's 8 slots of its case's row

        // ---- pass 1: largest squared distance
        double max_d2 = 0.0;
        for (int ch = 0; ch < nchunks; ++ch) {
            stage(ch * CH);
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                const int k = ch * CH + h * SPL + s;
                double d2 = 0.0;
#pragma unroll
                for (int m = 0; m < DIM; ++m) { const double dd = xr[s * DIM + m] - xi[m]; d2 = fma(dd, dd, d2); }
                d2 = (k < nkc) ? d2 : 0.0;
                max_d2 = d2 > max_d2 ? d2 : max_d2;
            }
        }
#pragma unroll
        for (int off = TC; off < WV; off <<= 1) { const double o = __shfl_xor(max_d2, off, WV); max_d2 = o > max_d2 ? o : max_d2; }
        const double inv_max = inverse_max(max_d2);

        // ---- pass 2: moments
        double mu[NM], nu[NO];
#pragma unroll
        for (int e = 0; e < NM; ++e) mu[e] = 0.0;
#pragma unroll
        for (int a = 0; a < NO; ++a) nu[a] = 0.0;
        for (int ch = 0; ch < nchunks; ++ch) {
            // this lane's fk values of the chunk, straight from global memory (8 contiguous doubles; slots beyond the row are masked)
            double f[SPL];
            {
                const int kb = ch * CH + h * SPL;
                const double* gr = p.fk + jc * (long long)K;
#pragma unroll
                for (int s = 0; s < SPL; s += 2) {
                    const int kq = (kb + s < K) ? kb + s : K - 2;
                    const cd2_ v = *reinterpret_cast<const cd2_*>(gr + kq);
                    f[s] = v.x; f[s + 1] = v.y;
                }
            }
            stage(ch * CH, cache_x);
            if constexpr (ITER) {
                if (cache_x) {
#pragma unroll
                    for (int s = 0; s < SPL; ++s) { const int k = ch * CH + h * SPL + s; if (k < K) sFall[c * RFA + k] = f[s]; }
                }
            }
#pragma unroll 2
            for (int s = 0; s < SPL; ++s) {
                const int k = ch * CH + h * SPL + s;
                const bool live = k < nkc;
                double d[DIM];
#pragma unroll
                for (int m = 0; m < DIM; ++m) { const double dd = xr[s * DIM + m] - xi[m]; d[m] = live ? dd : 0.0; }
                double d2 = 0.0;
#pragma unroll
                for (int m = 0; m < DIM; ++m) d2 = fma(d[m], d[m], d2);
                const double w = live ? weight(d2, inv_max, uniform) : 0.0;
                accumulate_moments_best<DIM, ORDER>(mu, nu, d, w, live ? f[s] : 0.0);
            }
        }
#pragma unroll
        for (int off = TC; off < WV; off <<= 1) {
#pragma unroll
            for (int e = 0; e < NM; ++e) mu[e] += __shfl_xor(mu[e], off, WV);
#pragma unroll
            for (int a = 0; a < NO; ++a) nu[a] += __shfl_xor(nu[a], off, WV);
        }
        constexpr unsigned long long FULL = (1ull << NO) - 1ull;
        if constexpr (ITER) {
            double* fio = p.fi + jc * p.sfi_j;
            double M[NE], g[NO], val[NO];
            expand_moments<DIM, ORDER>(mu, nu, M, g);
#pragma unroll
            for (int a = 0; a < NO; ++a) val[a] = (((known & ~dropped) >> a) & 1ull) ? fio[a] : 0.0;
            eliminate_knowns<NO>(M, g, known, val);
            ldlt_factor<NO>(M);
            ldlt_solve<NO>(M, g);
            double fi[NO];
#pragma unroll
            for (int a = 0; a < NO; ++a) fi[a] = ((known >> a) & 1ull) ? (((dropped >> a) & 1ull) ? fio[a] : val[a]) : g[a];
            bool done = !(valid && known != FULL);
            bool broke = false;
            int it_case = 0;
            double prev_norm = -1.0;
            for (int it = 0; it < p.max_iter; ++it) {
                if (__ballot(!done) == 0ull) break;                              // every case of the tile has stopped
                double norm = 0.0, r[NO];
#pragma unroll
                for (int a = 0; a < NO; ++a) r[a] = 0.0;
                auto slot = [&](int k, const double* xs, double fv) {
                    const bool live = k < nkc;
                    double d[DIM], cc[NO];
#pragma unroll
                    for (int m = 0; m < DIM; ++m) { const double dd = xs[m] - xi[m]; d[m] = live ? dd : 0.0; }
                    const double d2 = monomials<DIM, ORDER>(d, cc);
                    const double w = live ? weight(d2, inv_max, uniform) : 0.0;
                    double model = fi[0];                                        // taylor_*D (polyeval.pyx): sum_a c[a] fi[a]
#pragma unroll
                    for (int a = 1; a < NO; ++a) model = fma(cc[a], fi[a], model);
                    const double res = live ? fv - model : 0.0;
                    const double ar = fabs(res);
                    norm = ar > norm ? ar : norm;                                // impl.pyx:1037-1041
                    const double wr = w * res;
#pragma unroll
                    for (int a = 0; a < NO; ++a) r[a] = fma(wr, (a == 0) ? 1.0 : cc[a], r[a]);
                };
                if (cache_x) {
                    // the tile is in LDS: this lane's slots of every chunk, no global load, no barrier
                    for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll 2
                        for (int s = 0; s < SPL; ++s) {
                            const int k = ch * CH + h * SPL + s, kk = k < K ? k : K - 1;
                            slot(k, sXall + c * RSA + kk * DIM, sFall[c * RFA + kk]);
                        }
                    }
                } else
                for (int ch = 0; ch < nchunks; ++ch) {
                    double f[SPL];
                    {
                        const int kb = ch * CH + h * SPL;
                        const double* gr = p.fk + jc * (long long)K;
#pragma unroll
                        for (int s = 0; s < SPL; s += 2) {
                            const int kq = (kb + s < K) ? kb + s : K - 2;
                            const cd2_ v = *reinterpret_cast<const cd2_*>(gr + kq);
                            f[s] = v.x; f[s + 1] = v.y;
                        }
                    }
                    stage(ch * CH);
#pragma unroll 2
                    for (int s = 0; s < SPL; ++s) slot(ch * CH + h * SPL + s, xr + s * DIM, f[s]);
                }
#pragma unroll
                for (int off = TC; off < WV; off <<= 1) {
                    const double o = __shfl_xor(norm, off, WV); norm = o > norm ? o : norm;
#pragma unroll
                    for (int a = 0; a < NO; ++a) r[a] += __shfl_xor(r[a], off, WV);
                }
                if (!done) {
                    if (norm == prev_norm) { broke = true; done = true; it_case = it; }      // impl.pyx:1057
                    else {
                        prev_norm = norm;
#pragma unroll
                        for (int a = 0; a < NO; ++a) if ((known >> a) & 1ull) r[a] = 0.0;    // knowns of the correction are 0
                        ldlt_solve<NO>(M, r);
#pragma unroll
                        for (int a = 0; a < NO; ++a) if (!((known >> a) & 1ull)) fi[a] += r[a];
                    }
                }
            }
            if (valid && h == 0 && known != FULL) {
#pragma unroll
                for (int a = 0; a < NO; ++a)
                    if (!((known >> a) & 1ull)) fio[a] = fi[a];
                const int iters = broke ? it_case : (p.max_iter > 0 ? p.max_iter : 1);       // for/else, impl.pyx:1080-1081
                if (p.iters_out) atomicMax(p.iters_out, iters);
            }
        } else
        if constexpr (INV) {
            double* fio = p.fi + jc * p.sfi_j;
            double M[NE], rhs[NO];
            expand_moments<DIM, ORDER>(mu, nu, M, rhs);
            {
                double val[NO];
#pragma unroll
                for (int a = 0; a < NO; ++a) val[a] = (((known & ~dropped) >> a) & 1ull) ? fio[a] : 0.0;
                eliminate_knowns<NO>(M, rhs, known, val);
            }
            ldlt_factor<NO>(M);
            ldlt_solve<NO>(M, rhs);
            if (valid && known != FULL) {
                if (h == 0) {
#pragma unroll
                    for (int a = 0; a < NO; ++a)
                        if (!((known >> a) & 1ull)) fio[a] = rhs[a];
                }
                double* wi = p.ws + j * (long long)(NO * NO);
#pragma unroll 1
                for (int col = h; col < NO; col += LPC) {
                    double sv[NO];
                    const bool kcol = (known >> col) & 1ull;           // known: zero row and column
#pragma unroll
                    for (int a = 0; a < NO; ++a) sv[a] = (a == col && !kcol) ? 1.0 : 0.0;
                    ldlt_solve<NO>(M, sv);
#pragma unroll
                    for (int a = 0; a < NO; ++a) wi[col * NO + a] = sv[a];
                }
            }
        } else
        if (valid && h == 0 && known != FULL) {
            double* fio = p.fi + j * p.sfi_j;
            double M[NE], rhs[NO];
            expand_moments<DIM, ORDER>(mu, nu, M, rhs);
            if (known) {
                double val[NO];
#pragma unroll
                for (int a = 0; a < NO; ++a) val[a] = (((known & ~dropped) >> a) & 1ull) ? fio[a] : 0.0;
                eliminate_knowns<NO>(M, rhs, known, val);
            }
            ldlt_factor<NO>(M);
            ldlt_solve<NO>(M, rhs);
#pragma unroll
            for (int a = 0; a < NO; ++a)
                if (!((known >> a) & 1ull)) fio[a] = rhs[a];
        }
    }
}

template <int DIM, int ORDER, bool INV = false, bool ITER = false>
static int launch_chunk(const KParams& p, long long K, hipStream_t stream) {
    constexpr int MINW = (ndofs(DIM, ORDER) > ((INV || ITER) ? 6 : 10)) ? 1 : 2;
    const long long ntiles = (p.ncases + 15) / 16;
    static KernelSetup setup;
    auto kern = fit_chunk_kernel<DIM, ORDER, MINW, INV, ITER>;
    // refinement: the tile's xk and fk rows stay in LDS between the sweeps when four such waves still fit one CU
    const size_t cache_bytes = (size_t)16 * (((K * DIM) | 1) + (K | 1)) * sizeof(double);
    const size_t fixed_bytes = (size_t)(16 * (32 * DIM + 2) + 2) * sizeof(double);
    const bool cache_x = ITER && 4 * (cache_bytes + fixed_bytes + 256) <= 160 * 1024;
    const size_t dyn = cache_x ? cache_bytes : 0;
    // (the grid is sized for the workgroups that can really be resident WITH this launch's dynamic LDS: ADVICE r3 — it was asked for
    // with 0 bytes, i.e. for more workgroups than co-reside when the refinement caches its rows; the occupancy depends on K then, so
    // it is not memoised)
    long long grid = 0;
    int rc = persistent_grid(reinterpret_cast<const void*>(kern), 64, dyn, 0, dyn == 0, setup, &grid);
    if (rc != WLSQM_OK) return rc;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), dyn, stream, p, ntiles, (int)K, cache_x);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel(ITER ? "chunk-refine" : INV ? "chunk-inverse" : "chunk");
    return WLSQM_OK;
}

static bool chunk_layout_ok(int dimension, const KParams& p, long long K) {
    if (p.hoods || p.case_index || K < 2 || (K % 2) != 0 || K > 0x3fffffff) return false;
    if (p.sxk_k != dimension || p.sxk_j != K * dimension || p.sfk_k != 1 || p.sfk_j != K) return false;
    return ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(p.fk)) & 15u) == 0;
}

// Dense contiguous basic fits of any K (even, 16-byte aligned rows) with at most 15 unknowns.
int launch_fit_chunk(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    if (p.do_sens || p.iterative || !chunk_layout_ok(dimension, p, K)) return WLSQM_OK;
#define CCASE(D, O) if (dimension == D && order == O) { *handled = true; return launch_chunk<D, O>(p, K, stream); }
    CCASE(1, 0) CCASE(1, 1) CCASE(1, 2) CCASE(1, 3) CCASE(1, 4)
    CCASE(2, 0) CCASE(2, 1) CCASE(2, 2) CCASE(2, 3) CCASE(2, 4)
    CCASE(3, 0) CCASE(3, 1) CCASE(3, 2)
#undef CCASE
    return WLSQM_OK;
}

// Fit + iterative refinement in one kernel (ITER above) for dense contiguous rows of any even K: what api.hip sends here are the
// shapes no tile kernel with extras takes (2D orders 3-4: K > 64 / every K), before the lane-per-case kernel.
int launch_fit_chunk_refine(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    const char* no_ = getenv("WLSQM_HIP_DISABLE_CHUNK_REFINE");                 // A/B against the lane kernel
    if (no_ && no_[0] == '1') return WLSQM_OK;
    if (!p.iterative || p.do_sens || !chunk_layout_ok(dimension, p, K)) return WLSQM_OK;
#define CCASE(D, O) if (dimension == D && order == O) { *handled = true; return launch_chunk<D, O, false, true>(p, K, stream); }
    CCASE(1, 0) CCASE(1, 1) CCASE(1, 2) CCASE(1, 3) CCASE(1, 4)
    CCASE(2, 0) CCASE(2, 1) CCASE(2, 2) CCASE(2, 3) CCASE(2, 4)
    CCASE(3, 0) CCASE(3, 1) CCASE(3, 2)
#undef CCASE
    return WLSQM_OK;
}

// The same fit, which also leaves every case's inverse normal matrix at p.ws[j][no][no] (first kernel of fit_sens.hip).
bool chunk_inverse_ok(int dimension, int order, const KParams& p, long long K) {
    return ndofs(dimension, order) <= 15 && chunk_layout_ok(dimension, p, K);
}
int launch_fit_chunk_inverse(int dimension, int order, const KParams& p, long long K, hipStream_t stream) {
#define CCASE(D, O) if (dimension == D && order == O) return launch_chunk<D, O, true>(p, K, stream);
    CCASE(1, 0) CCASE(1, 1) CCASE(1, 2) CCASE(1, 3) CCASE(1, 4)
    CCASE(2, 0) CCASE(2, 1) CCASE(2, 2) CCASE(2, 3) CCASE(2, 4)
    CCASE(3, 0) CCASE(3, 1) CCASE(3, 2)
#undef CCASE
    set_error("fit_chunk_inverse: unsupported (dimension, order)");
    return WLSQM_EVALUE;
}

}  // namespace wlsqm
