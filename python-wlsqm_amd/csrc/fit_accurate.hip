// fit_accurate.hip — the ACCURATE numerics mode (WLSQM_HIP_STRICT=2 / wlsqm_hip_set_strict(2)) for gfx950: the mode that is meant
// to satisfy BOTH halves of the north-star target at once — derivative DOFs within 1e-10 of the reference's on every column AND a
// fit rate at the HBM roofline's scale.
//
// What it is.  The strict mode (fit_strict.hip) replays the reference's floating-point operations one for one and is 5x slower
// than the fast kernels; the fast kernels are 2.4e-10 from the reference on configs[1].  profiles/r03_attribution.txt says which of
// the fast kernels' choices cost that distance: the split neighbour sums, the reciprocal in the weights, the unscaled LDL^T — and
// that ONE choice is free: assembling only the upper triangle of the normal matrix and mirroring it (3.33e-11 against the strict
// mode's 3.36e-11 on configs[1], 1.71e-11 against 1.69e-11 on configs[4]; oracle/variants.c V_SYM).  This kernel is therefore the
// reference's arithmetic with exactly that one change:
//
//   make_c_{2,3}D        impl.pyx:286-432, 70-269     the reference's grouping of every scaled monomial
//   Case_make_weights    infra.pyx:668-702            CORRECTLY ROUNDED quotient and root (bit-identical to IEEE / and sqrt)
//   make_A               impl.pyx:566-602             entry (j, m), m >= j: sum_k (w c_m) c_j, k ascending in ONE lane, no FMA;
//                                                     entry (m, j) := entry (j, m)   <- the one change
//   rescale_ruiz2001_c   lapackdrivers.pyx:553-623    the same sweeps with the same stop test; on a symmetric matrix the row and the
//                                                     column pass see the same numbers (DR == DC bit for bit), so one of them is run
//   dgetrf / dgetrs      lapackdrivers.pyx:1628-1665  unblocked partial-pivot LU, first maximum wins
//   solve                impl.pyx:731-846             right-hand side sums, un-scaling
//
// and its output is BIT-IDENTICAL to oracle/variants.c with V_SYM (tests/test_gpu_accurate.py), layout- and tile-mate-independent.
//
// Where the time of the strict register kernel went, and what is different here (same bits, fewer instructions):
//   * IEEE divide = v_div_scale x2 + v_rcp + 4 fma + mul + fma + v_div_fmas + v_div_fixup (11 instructions; read from the ISA).
//     For operands whose exponents are far from the ends of the range the two scale instructions return their inputs, div_fmas is a
//     plain fma and div_fixup the identity: the remaining 8 instructions ARE the quotient (fdiv below; same operations, same
//     bits).  The weights divide every squared distance by the same max_d2: the refined reciprocal is computed once per case and a
//     quotient costs 3 instructions.  IEEE sqrt likewise: 17 -> 10 (no scaling, no class test).  Whether a case's operands are in
//     the safe range is CHECKED (squared distances, matrix entries, the running scale factors); a wave with any case outside it
//     runs the same code with the compiler's IEEE sequences instead (same bits where both apply) — never a silent approximation.
//     tools/ubench/exact_div_sqrt.hip holds the two sequences against `/` and sqrt() on 2^31 random operand pairs.
//   * the matrix is 21 / 55 sums instead of 36 / 100, the equilibration 21 / 55 quotients per sweep instead of 36 / 100 and one
//     set of roots and scale updates instead of two.
//   * the row exchanges of the LU (selects over the candidate rows: registers cannot be indexed by a lane's pivot row) are skipped
//     by a wave whose 64 cases all keep the diagonal pivot in that column — 99.9 % of configs[1]'s columns, 99.5 % of configs[4]'s.
//   * the neighbour rows of the 64 cases of a wave reach their lanes through LDS in chunks of 8 neighbours: global loads are
//     coalesced 16-byte pieces of whole 128- / 192-byte runs (the strict register kernel's lanes each read their own row: 64 cache
//     lines per load instruction), the next chunk is in flight in registers while the current one is consumed.
// One lane per case; a wave owns 64 consecutive cases.  Cases with a known DOF, sensitivities, refinement, systems above 10
// unknowns and 1D fits are NOT taken here: in accurate mode they run the strict kernels (bit-identical to the oracle, i.e. at least
// as close to the reference).  Which kernel takes a case depends on that case alone.
#include <atomic>
#include <type_traits>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_strict.hpp"

#pragma clang fp contract(off)      // the reference is gcc -O2 on x86-64: no contraction; every fma() below is spelled out

#ifndef WLSQM_ACC_MINW6
#define WLSQM_ACC_MINW6 2           // waves per SIMD the kernel of the systems up to 6 unknowns is compiled for
#endif
#ifndef WLSQM_ACC_MINW10
#define WLSQM_ACC_MINW10 1          // ... of the 10-unknown systems
#endif

namespace wlsqm {

namespace acc {

typedef double d2_ __attribute__((ext_vector_type(2)));

// ---- correctly rounded quotient and root for in-range operands: the instruction sequences hipcc emits for `/` and sqrt()
// (LLVM AMDGPU LowerFDIV64 / lowerFSQRTF64) without the range scaling and the special-case fix-up
__device__ __forceinline__ double rcp_refined(double b) {          // v_rcp_f64 + two Newton steps
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0); r = fma(r, e, r);
    e = fma(-b, r, 1.0); r = fma(r, e, r);
    return r;
}
__device__ __forceinline__ double div_by(double a, double b, double r) {   // r = rcp_refined(b)
    const double q = a * r;
    const double e = fma(-b, q, a);
    return fma(e, r, q);
}
struct FastOps {
    static __device__ __forceinline__ double div(double a, double b) { return div_by(a, b, rcp_refined(b)); }
    static __device__ __forceinline__ double rcp_of(double b) { return rcp_refined(b); }
    static __device__ __forceinline__ double div_r(double a, double b, double r) { return div_by(a, b, r); }
    static __device__ __forceinline__ double sqrt(double x) {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = y * 0.5;
        const double r = fma(-h, g, 0.5);
        g = fma(g, r, g); h = fma(h, r, h);
        double d = fma(-g, g, x); g = fma(d, h, g);
        d = fma(-g, g, x); g = fma(d, h, g);
        return g;
    }
};
struct IeeeOps {                                                     // the compiler's sequences: any operand
    static __device__ __forceinline__ double div(double a, double b) { return a / b; }
    static __device__ __forceinline__ double rcp_of(double) { return 0.; }
    static __device__ __forceinline__ double div_r(double a, double b, double) { return a / b; }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
};

// safe range of the fast sequences (see the header): every nonzero magnitude that enters a fast quotient or root lies in
// [2^-200, 2^200] and every running scale factor in [2^-140, 2^140]: exponent differences stay below 768, no operand or result
// is subnormal, no numerator has a biased exponent <= 53 — the conditions under which v_div_scale / v_div_fixup are the identity
constexpr double RANGE_LO = 0x1p-200, RANGE_HI = 0x1p200, SCALE_LO = 0x1p-140, SCALE_HI = 0x1p140;

constexpr int CH = 8;                                                // neighbours per staged chunk

template <int N> __host__ __device__ constexpr int utri(int i, int m) { return i * N - i * (i - 1) / 2 + (m - i); }   // i <= m < N

__host__ __device__ constexpr int minw(int NO) { return NO <= 6 ? WLSQM_ACC_MINW6 : WLSQM_ACC_MINW10; }

}  // namespace acc

// DENSE: contiguous rows xk[ncases][K][DIM], fk[ncases][K] with 16-byte aligned bases and rows (staged through LDS); otherwise the
// rows are read per lane through strict::Rows (any strides, index-based input, order buckets) — the same arithmetic, the same bits.
template <int DIM, int ORDER, bool DENSE>
__global__ __launch_bounds__(64, acc::minw(ndofs(DIM, ORDER))) void fit_accurate_kernel(const KParams p) {
    using namespace strict;
    using namespace acc;
    constexpr int N = ndofs(DIM, ORDER), NE = N * (N + 1) / 2;
    constexpr int XPC = CH * DIM * 8 / 16, FPC = CH * 8 / 16;        // 16-byte pieces of one case's chunk: coordinates, values
    constexpr int XPITCH = CH * DIM + 2, FPITCH = CH + 2;            // doubles per staged row (+ 16 bytes: conflict-free b128 reads)
    __shared__ __attribute__((aligned(16))) double xs[DENSE ? 64 * XPITCH : 2];
    __shared__ __attribute__((aligned(16))) double fs[DENSE ? 64 * FPITCH : 2];

    const long long ncases = live_cases(p);
    const int lane = threadIdx.x;
    const long long t0 = (long long)blockIdx.x * 64, t = t0 + lane;
    const bool in_batch = t < ncases;
    const long long j = in_batch ? (p.case_index ? p.case_index[t] : t) : 0;
    // a case is taken here iff it has no known DOF (wave-mates do not matter: per case); p.do_sens / p.iterative never reach this kernel
    const bool active = in_batch && p.knowns[j * p.sknowns] == 0;
    if (!__any(active)) return;
    const int K = (int)p.max_nk;
    const int nk = active ? min(p.nk[j * p.snk], K) : 0;
    const bool uniform = active ? (p.wm[j * p.swm] == WLSQM_WEIGHT_UNIFORM) : true;
    double xi[DIM];
    Rows<DIM> rows{};
    if constexpr (DENSE) {
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = in_batch ? p.xi[j * p.sxi_j + m] : 0.;
    } else {
        if (p.hoods) {
            const long long pj = p.pidx ? p.pidx[j] : j;
#pragma unroll
            for (int m = 0; m < DIM; ++m) xi[m] = in_batch ? p.S[pj * DIM + m] : 0.;
            rows = Rows<DIM>{nullptr, 0, nullptr, 0, p.hoods + j * p.shoods_j, p.S, p.F};
        } else {
#pragma unroll
            for (int m = 0; m < DIM; ++m) xi[m] = in_batch ? p.xi[j * p.sxi_j + m] : 0.;
            rows = Rows<DIM>{p.xk + j * p.sxk_j, p.sxk_k, p.fk + j * p.sfk_j, p.sfk_k, nullptr, nullptr, nullptr};
        }
    }

    // ---- one pass over the neighbours of the wave's cases: consume(k, live, d, f) per lane, k ascending.  DENSE: chunks of CH
    // neighbours through LDS, the next chunk in flight in registers.  MASKED = false: every active lane has nk == K (wave-uniform).
    const int Q = (K + CH - 1) / CH;
    const int nvalid = (ncases - t0 < 64) ? (int)(ncases - t0) : 64;
    d2_ xr[DENSE ? XPC : 1], fr[DENSE ? FPC : 1];
    auto fetch = [&](int q, bool want_f) {                            // global -> registers, coalesced 16-byte pieces
        if constexpr (DENSE) {
            const char* xb = reinterpret_cast<const char*>(p.xk + t0 * (long long)K * DIM) + (size_t)q * (CH * DIM * 8);
            const int rowb = K * DIM * 8 - q * (CH * DIM * 8);        // bytes of a row from this chunk on
#pragma unroll
            for (int i = 0; i < XPC; ++i) {
                const int pi = i * 64 + lane, cc = pi / XPC, sub = pi - cc * XPC;
                if (cc < nvalid && sub * 16 < rowb)
                    xr[i] = *reinterpret_cast<const d2_*>(xb + (size_t)cc * ((size_t)K * DIM * 8) + sub * 16);
            }
            if (want_f) {
                const char* fb = reinterpret_cast<const char*>(p.fk + t0 * (long long)K) + (size_t)q * (CH * 8);
                const int frow = K * 8 - q * (CH * 8);
#pragma unroll
                for (int i = 0; i < FPC; ++i) {
                    const int pi = i * 64 + lane, cc = pi / FPC, sub = pi - cc * FPC;
                    if (cc < nvalid && sub * 16 < frow)
                        fr[i] = *reinterpret_cast<const d2_*>(fb + (size_t)cc * ((size_t)K * 8) + sub * 16);
                }
            }
        }
    };
    auto park = [&](bool want_f) {                                    // registers -> LDS rows
        if constexpr (DENSE) {
#pragma unroll
            for (int i = 0; i < XPC; ++i) {
                const int pi = i * 64 + lane, cc = pi / XPC, sub = pi - cc * XPC;
                *reinterpret_cast<d2_*>(xs + cc * XPITCH + sub * 2) = xr[i];
            }
            if (want_f) {
#pragma unroll
                for (int i = 0; i < FPC; ++i) {
                    const int pi = i * 64 + lane, cc = pi / FPC, sub = pi - cc * FPC;
                    *reinterpret_cast<d2_*>(fs + cc * FPITCH + sub * 2) = fr[i];
                }
            }
        }
    };
    // (the staged passes run back to back: pass P's first chunk is requested under pass P - 1's last)
    auto run_pass = [&](auto masked_tag, bool want_f, bool prefetched, bool more_passes, bool next_want_f, auto&& consume) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        if constexpr (DENSE) {
            if (!prefetched) fetch(0, want_f);
            for (int q = 0; q < Q; ++q) {
                __syncthreads();                                      // the previous chunk has been read by every lane
                park(want_f);
                __syncthreads();
                if (q + 1 < Q) fetch(q + 1, want_f);
                else if (more_passes) fetch(0, next_want_f);
                const double* xrow = xs + lane * XPITCH;
                const double* frow = fs + lane * FPITCH;
#pragma unroll
                for (int kk = 0; kk < CH; ++kk) {
                    const int k = q * CH + kk;
                    if (k < K) {                                      // wave-uniform
                        const bool live = MASKED ? (k < nk) : true;
                        double d[DIM];
#pragma unroll
                        for (int m = 0; m < DIM; ++m) { d[m] = xrow[kk * DIM + m] - xi[m]; if (MASKED) d[m] = live ? d[m] : 0.; }
                        double f = want_f ? frow[kk] : 0.;
                        if (MASKED) f = live ? f : 0.;
                        consume(k, live, d, f);
                    }
                }
            }
        } else {
            for (int k = 0; k < nk; ++k) {
                double d[DIM];
                rows.offset(k, xi, d);
                consume(k, true, d, want_f ? rows.value(k) : 0.);
            }
        }
    };
    const bool wave_full = __all(!active || nk == K);
    auto pass = [&](bool want_f, bool prefetched, bool more_passes, bool next_want_f, auto&& consume) {
        if (wave_full) run_pass(std::false_type{}, want_f, prefetched, more_passes, next_want_f, consume);
        else run_pass(std::true_type{}, want_f, prefetched, more_passes, next_want_f, consume);
    };

    // ---- pass 1 (make_c_nD, first half): the largest squared distance; the smallest too, for the range check of the fast weights
    double max_d2 = 0., min_d2 = RANGE_HI;
    pass(false, false, true, true, [&](int, bool live, const double (&d)[DIM], double) {
        double c[N];
        const double d2 = make_c<DIM, ORDER>(d, c);
        if (live) { if (d2 > max_d2) max_d2 = d2; if (!(d2 >= min_d2)) min_d2 = d2; }      // (a NaN distance lands in min_d2)
    });
    // fast weights: every squared distance of the case in the safe range (a neighbour AT the centre, a NaN or an empty
    // neighbourhood fail it and take the IEEE sequences); uniform weighting computes no quotient at all
    const bool w_ok = !active || uniform || (min_d2 >= RANGE_LO && max_d2 <= RANGE_HI && nk > 0);
    const bool fast_w = __all(w_ok);

    // ---- pass 2: make_A (upper triangle) and the right-hand side sums of solve (impl.pyx:768-787), k ascending
    double U[NE], b[N];
#pragma unroll
    for (int e = 0; e < NE; ++e) U[e] = 0.;
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] = 0.;
    auto accumulate = [&](auto ops_tag) {
        using OPS = decltype(ops_tag);
        const double rmax = OPS::rcp_of(max_d2);
        pass(true, true, false, false, [&](int, bool live, const double (&d)[DIM], double f) {
            double c[N];
            const double d2 = make_c<DIM, ORDER>(d, c);
            const double tmp = 1. - OPS::sqrt(OPS::div_r(d2, max_d2, rmax));
            double w = uniform ? 1. : weights_alpha + weights_beta * tmp * tmp;
            w = live ? w : 0.;
            const double wf = w * f;
#pragma unroll
            for (int om = 0; om < N; ++om) {
                const double wc = w * c[om];
#pragma unroll
                for (int oj = 0; oj <= om; ++oj) U[utri<N>(oj, om)] += wc * c[oj];
            }
#pragma unroll
            for (int oj = 0; oj < N; ++oj) b[oj] += wf * c[oj];
        });
    };
    if (fast_w) accumulate(FastOps{}); else accumulate(IeeeOps{});
    if (!active) return;                                              // (no barrier below this line)

    // ---- rescale_ruiz2001_c (lapackdrivers.pyx:553-623) on the symmetric matrix: DR == DC, DRp == DCp, rs == cs bit for bit
    double rs[N], DRp[N];
    bool r_ok = true;
    {   // range check of the fast sweeps: every nonzero entry in the safe range and no zero row
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double rowmax = 0.;
#pragma unroll
            for (int m = 0; m < N; ++m) {
                const double a = fabs(U[i <= m ? utri<N>(i, m) : utri<N>(m, i)]);
                r_ok = r_ok && (a == 0. || (a >= RANGE_LO && a <= RANGE_HI));
                rowmax = a > rowmax ? a : rowmax;
            }
            r_ok = r_ok && rowmax >= RANGE_LO;
        }
    }
    auto ruiz = [&](auto ops_tag) -> bool {
        using OPS = decltype(ops_tag);
        double slo = 1., shi = 1.;
#pragma unroll
        for (int i = 0; i < N; ++i) { rs[i] = 1.; DRp[i] = 1.; }
        for (int it = 0; it < 100; ++it) {
            double DR[N];
#pragma unroll
            for (int i = 0; i < N; ++i) DR[i] = 0.;
#pragma unroll
            for (int m = 0; m < N; ++m) {
#pragma unroll
                for (int i = 0; i <= m; ++i) {
                    const double q = fabs(OPS::div(U[utri<N>(i, m)], DRp[i] * DRp[m]));
                    if (q > DR[i]) DR[i] = q;
                    if (i != m) { if (q > DR[m]) DR[m] = q; }
                }
            }
            double acc = 0.;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const double s = OPS::sqrt(DR[i]);
                DRp[i] *= s; rs[i] = OPS::div(rs[i], s);
                slo = DRp[i] < slo ? DRp[i] : slo; shi = DRp[i] > shi ? DRp[i] : shi;
                const double tmp = fabs(1. - s * s);
                if (i == 0) acc = tmp; else if (tmp > acc) acc = tmp;
            }
            if (acc < ruiz_epsilon) break;                            // (the column test sees the same numbers)
        }
        return slo >= SCALE_LO && shi <= SCALE_HI;
    };
    bool fast_done = false;
    if (__all(r_ok)) { r_ok = ruiz(FastOps{}); fast_done = true; }
    if (!fast_done || !__all(r_ok)) (void)ruiz(IeeeOps{});           // IEEE sequences for the whole wave: the same bits where both apply

    // apply_scaling_c (lapackdrivers.pyx:293-299): A[i][m] *= rs[i] * cs[m] (commutative: the scaled matrix is symmetric too)
    double A[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int m = i; m < N; ++m) { const double v = U[utri<N>(i, m)] * (rs[i] * rs[m]); A[i][m] = v; A[m][i] = v; }

    // dgetrf (unblocked dgetf2 semantics).  The row exchange is written as selects over the candidate rows; a wave none of whose
    // cases leaves the diagonal pivot in this column skips it.
    int ipiv[N];
#pragma unroll
    for (int c0 = 0; c0 < N; ++c0) {
        int pv = c0; double best = fabs(A[c0][c0]), pval = A[c0][c0];
#pragma unroll
        for (int i = c0 + 1; i < N; ++i) { const double v = fabs(A[i][c0]); if (v > best) { best = v; pv = i; pval = A[i][c0]; } }
        ipiv[c0] = pv;
        if (__any(pv != c0)) {
#pragma unroll
            for (int i = c0 + 1; i < N; ++i) {
                const bool sw = (pv == i);
#pragma unroll
                for (int m = 0; m < N; ++m) { const double u = A[c0][m], v = A[i][m]; A[c0][m] = sw ? v : u; A[i][m] = sw ? u : v; }
            }
        }
        if (pval != 0.) {
            const double r = 1. / A[c0][c0];
#pragma unroll
            for (int i = c0 + 1; i < N; ++i) A[i][c0] *= r;
        }
#pragma unroll
        for (int m = c0 + 1; m < N; ++m) {
            const double u = A[c0][m];
#pragma unroll
            for (int i = c0 + 1; i < N; ++i) A[i][m] -= A[i][c0] * u;
        }
    }
    // solve (impl.pyx:731-846) without knowns: b = row_scale * sums, dgetrs('N'), un-scale
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] = rs[i] * b[i];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (__any(ipiv[i] != i)) {
#pragma unroll
            for (int q = i + 1; q < N; ++q) { const bool sw = (ipiv[i] == q); const double u = b[i], v = b[q]; b[i] = sw ? v : u; b[q] = sw ? u : v; }
        }
    }
#pragma unroll
    for (int c0 = 0; c0 < N; ++c0)
#pragma unroll
        for (int i = c0 + 1; i < N; ++i) b[i] -= A[i][c0] * b[c0];
#pragma unroll
    for (int c0 = N - 1; c0 >= 0; --c0) {
        b[c0] /= A[c0][c0];
#pragma unroll
        for (int i = 0; i < c0; ++i) b[i] -= A[i][c0] * b[c0];
    }
    double* const fio = p.fi + j * p.sfi_j;
#pragma unroll
    for (int i = 0; i < N; ++i) fio[i] = b[i] * rs[i];
}

template <int DIM, int ORDER>
static int launch_accurate(const KParams& p, hipStream_t stream) {
    const long long groups = (p.ncases + 63) / 64;
    if (groups <= 0) return WLSQM_OK;
    if (groups > 0x7fffffffLL) { set_error("too many cases for one launch"); return WLSQM_EVALUE; }
    const long long K = p.max_nk;
    const bool dense = !p.hoods && !p.case_index && p.xk && p.fk && K >= 2 && K % 2 == 0 && p.sxk_k == DIM && p.sxk_j == K * DIM &&
                       p.sfk_k == 1 && p.sfk_j == K && ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(p.fk)) & 15u) == 0 &&
                       !getenv("WLSQM_HIP_ACCURATE_NO_STAGE");
    if (dense) hipLaunchKernelGGL((fit_accurate_kernel<DIM, ORDER, true>), dim3((unsigned)groups), dim3(64), 0, stream, p);
    else hipLaunchKernelGGL((fit_accurate_kernel<DIM, ORDER, false>), dim3((unsigned)groups), dim3(64), 0, stream, p);
    WLSQM_HIP_CHECK(hipGetLastError());
    return WLSQM_OK;
}

// Accurate mode, basic fits of the 2D / 3D systems up to 10 unknowns: every case WITHOUT a known DOF is fitted here (the strict
// kernels, launched behind this one by launch_fit_strict, leave exactly those cases alone).  *handled = false: the shape has no
// accurate kernel (1D, more than 10 unknowns) and the strict kernels take every case.
int launch_fit_accurate(int dimension, int order, const KParams& p, hipStream_t stream, bool* handled) {
    *handled = false;
    if (p.do_sens || p.iterative) return WLSQM_OK;
#define CASE(D, O) if (dimension == D && order == O) { *handled = true; return launch_accurate<D, O>(p, stream); }
    CASE(2, 0) CASE(2, 1) CASE(2, 2) CASE(2, 3)
    CASE(3, 0) CASE(3, 1) CASE(3, 2)
#undef CASE
    return WLSQM_OK;
}

}  // namespace wlsqm
