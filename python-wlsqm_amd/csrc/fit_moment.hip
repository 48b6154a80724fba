// fit_moment.hip — high-order 2D fast path (order 3 and 4): moment-based assembly.
//
// For order 4 the normal matrix has 120 unique entries plus 15 right-hand-side entries; keeping them
// in VGPRs (fit_lane / fit_tile) costs ~270 registers per lane, i.e. one wave per SIMD, and every
// tile variant measured slower than the generic kernel.  But
//     M[a,b] = sum_k w_k c_a(k) c_b(k) = mu(P_a + P_b) / (P_a! P_b!),   mu(P) = sum_k w_k dx^p dy^q,
//     g[a]   = sum_k w_k f_k c_a(k)    = nu(P_a) / P_a!,                nu(P) = sum_k w_k f_k dx^p dy^q
// (c_a = dx^p dy^q / (p! q!), impl.pyx:331-349; M as impl.pyx:601, g as :774), so the 120 entries are
// 45 distinct moments of total degree <= 8.  Two kernels:
//   A. moment_tile_kernel: the LDS-tiled streaming pass of fit_tile.hip (same staging, same weights), but
//      each lane accumulates only the 45 + 15 moments with a multiply chain (t(p,q) = t(p-1,q) dx or
//      t(p,q-1) dy): ~95 fp64 operations per neighbour instead of ~150, 60 accumulators instead of 135.
//      The reduced moments go to a structure-of-arrays workspace in HBM (480 B per case).
//   B. moment_solve_kernel: one lane per case expands M and g from the moments (compile-time factorial
//      constants), eliminates knowns, runs the in-register LDL^T and substitution and stores fi.
// The extra 960 B/case of workspace traffic is affordable because this configuration is fp64-VALU-bound.
#include <cstdlib>
#include <mutex>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

namespace wlsqm {

constexpr int MWV = 64;
typedef double dbl2_ __attribute__((ext_vector_type(2)));

// moments of a 2D fit of order O: multi-indices (p,q) with p+q <= 2*O, graded: idx = T(p+q) + q
__host__ __device__ constexpr int tri_num(int d) { return d * (d + 1) / 2; }
__host__ __device__ constexpr int midx(int p, int q) { return tri_num(p + q) + q; }
__host__ __device__ constexpr int nmoments(int order) { return tri_num(2 * order + 1); }
__host__ __device__ constexpr double inv_fact(int n) { return n == 0 ? 1.0 : n == 1 ? 1.0 : n == 2 ? 0.5 : n == 3 ? 1.0 / 6.0 : 1.0 / 24.0; }

// One neighbour into the moment accumulators.  mu: degrees 0..2*ORDER; nu: degrees 0..ORDER (DOF order, which for
// 2D IS the graded order idx = T(p+q)+q, defs.pyx:107-125).
template <int ORDER>
__device__ __forceinline__ void accumulate_moments(double (&mu)[nmoments(ORDER)], double (&nu)[ndofs(2, ORDER)],
                                                   double dx, double dy, double w, double f) {
    constexpr int D = 2 * ORDER;
    double prev[D + 1], cur[D + 1];
    prev[0] = w;
    mu[0] += w;
    nu[0] = fma(w, f, nu[0]);
#pragma unroll
    for (int d = 1; d <= D; ++d) {
#pragma unroll
        for (int q = 0; q <= d; ++q) {
            // t(d-q, q) = t(d-q-1, q) * dx  (q < d)   or   t(0, d) = t(0, d-1) * dy
            const double parent = (q < d) ? prev[q] : prev[q - 1];
            const double var = (q < d) ? dx : dy;
            if (d < D) {
                const double t = parent * var;
                cur[q] = t;
                mu[tri_num(d) + q] += t;
                if (d <= ORDER) nu[tri_num(d) + q] = fma(t, f, nu[tri_num(d) + q]);
            } else {
                mu[tri_num(d) + q] = fma(parent, var, mu[tri_num(d) + q]);
            }
        }
        if (d < D) {
#pragma unroll
            for (int q = 0; q <= d; ++q) prev[q] = cur[q];
        }
    }
}

__host__ __device__ constexpr int mround_up_mod(int v, int m, int r) { return v + ((r - v % m) % m + m) % m; }

template <int ORDER, int K, int KSPLIT, int LPC>
struct MomentGeom {
    static constexpr int NO = ndofs(2, ORDER), NM = nmoments(ORDER), NACC = NM + NO;
    static constexpr int TC = MWV / LPC, NT = MWV * KSPLIT, SHARES = KSPLIT * LPC, KPL = K / SHARES;
    static constexpr int RS = mround_up_mod(K * 2, 4, 2), FS = mround_up_mod(K, 2, 1);
    static constexpr int XCH = TC * K, FCH = TC * K / 2;           // 16-byte chunks per tile (xk: one (x,y) pair each)
    static constexpr int NX = (XCH + NT - 1) / NT, NF = (FCH + NT - 1) / NT;
    static constexpr int CPRX = K, CPRF = K / 2;
    static constexpr int LDS_TILE = TC * (RS + FS), LDS_RED = (KSPLIT - 1) * NACC * TC;
    static constexpr int LDS_MAIN = LDS_TILE > LDS_RED ? LDS_TILE : LDS_RED;
    static constexpr size_t LDS_BYTES = sizeof(double) * (LDS_MAIN + SHARES * TC);
    static_assert(K % SHARES == 0 && K % 2 == 0, "K must split evenly");
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
};

// Kernel A: streaming pass, moments to the SoA workspace ws[e * ncases + j].
template <int ORDER, int K, int KSPLIT, int LPC, int UNR, int MINW>
__global__ __launch_bounds__(MWV * KSPLIT, MINW) void moment_tile_kernel(const KParams p, const long long ntiles,
                                                                         double* __restrict__ ws) {
    using G = MomentGeom<ORDER, K, KSPLIT, LPC>;
    constexpr int NO = G::NO, NM = G::NM, NACC = G::NACC, TC = G::TC, NT = G::NT, RS = G::RS, FS = G::FS;
    constexpr int NX = G::NX, NF = G::NF, CPRX = G::CPRX, CPRF = G::CPRF, KPL = G::KPL;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sX = lds;
    double* sF = lds + TC * RS;
    double* sMax = lds + G::LDS_MAIN;

    const int tid = threadIdx.x, lane = tid & (MWV - 1), wave = tid / MWV;
    const int c = lane % TC, h = lane / TC, share = wave * LPC + h, k0 = share * KPL;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long j0 = tile * TC, j = j0 + c;
        const bool valid = j < p.ncases;
        const long long jc = valid ? j : p.ncases - 1;
        const long long nvalid = (p.ncases - j0 < TC) ? (p.ncases - j0) : TC;

        dbl2_ bx[NX], bf[NF];
        {
            const dbl2_* gx = reinterpret_cast<const dbl2_*>(p.xk + j0 * (long long)(K * 2));
            const dbl2_* gf = reinterpret_cast<const dbl2_*>(p.fk + j0 * (long long)K);
            const long long xlim = nvalid * CPRX, flim = nvalid * CPRF;
#pragma unroll
            for (int i = 0; i < NX; ++i) { const long long q = tid + (long long)i * NT; bx[i] = gx[q < xlim ? q : xlim - 1]; }
#pragma unroll
            for (int i = 0; i < NF; ++i) { const long long q = tid + (long long)i * NT; bf[i] = gf[q < flim ? q : flim - 1]; }
        }
        const int nkc = min(p.nk[jc * p.snk], K);
        const bool uniform = (p.wm[jc * p.swm] == WLSQM_WEIGHT_UNIFORM);
        const double xi0 = p.xi[jc * p.sxi_j], xi1 = p.xi[jc * p.sxi_j + 1];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int q = tid + i * NT;
            if (G::XCH % NT == 0 || q < G::XCH) {
                const int r = q / CPRX, c2 = q - r * CPRX;
                *reinterpret_cast<dbl2_*>(sX + r * RS + 2 * c2) = bx[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int q = tid + i * NT;
            if (G::FCH % NT == 0 || q < G::FCH) {
                const int r = q / CPRF, c2 = q - r * CPRF;
                double* d = sF + r * FS + 2 * c2;
                d[0] = bf[i].x; d[1] = bf[i].y;
            }
        }
        __syncthreads();

        const double* xr = sX + c * RS;
        const double* fr = sF + c * FS;

        double max_d2 = 0.0;
#pragma unroll UNR
        for (int kk = 0; kk < KPL; ++kk) {
            const int k = k0 + kk;
            const dbl2_ xy = *reinterpret_cast<const dbl2_*>(xr + 2 * k);
            const double dx = xy.x - xi0, dy = xy.y - xi1;
            double d2 = dx * dx + dy * dy;
            d2 = (k < nkc) ? d2 : 0.0;
            max_d2 = d2 > max_d2 ? d2 : max_d2;
        }
        sMax[share * TC + c] = max_d2;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < G::SHARES; ++s) { const double o = sMax[s * TC + c]; max_d2 = o > max_d2 ? o : max_d2; }
        const double inv_max = inverse_max(max_d2);

        double mu[NM], nu[NO];
#pragma unroll
        for (int e = 0; e < NM; ++e) mu[e] = 0.0;
#pragma unroll
        for (int a = 0; a < NO; ++a) nu[a] = 0.0;
#pragma unroll UNR
        for (int kk = 0; kk < KPL; ++kk) {
            const int k = k0 + kk;
            const bool live = k < nkc;
            const dbl2_ xy = *reinterpret_cast<const dbl2_*>(xr + 2 * k);
            const double dx = live ? xy.x - xi0 : 0.0, dy = live ? xy.y - xi1 : 0.0;
            const double d2 = dx * dx + dy * dy;
            const double w = live ? weight(d2, inv_max, uniform) : 0.0;
            accumulate_moments<ORDER>(mu, nu, dx, dy, w, live ? fr[k] : 0.0);
        }

        if constexpr (LPC > 1) {
#pragma unroll
            for (int off = TC; off < MWV; off <<= 1) {
#pragma unroll
                for (int e = 0; e < NM; ++e) mu[e] += __shfl_xor(mu[e], off, MWV);
#pragma unroll
                for (int a = 0; a < NO; ++a) nu[a] += __shfl_xor(nu[a], off, MWV);
            }
        }
        if constexpr (KSPLIT > 1) {
            __syncthreads();
            double* red = lds;
            if (wave > 0 && h == 0) {
                double* mine = red + (wave - 1) * (NACC * TC) + c;
#pragma unroll
                for (int e = 0; e < NM; ++e) mine[e * TC] = mu[e];
#pragma unroll
                for (int a = 0; a < NO; ++a) mine[(NM + a) * TC] = nu[a];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 1; w < KSPLIT; ++w) {
                    const double* other = red + (w - 1) * (NACC * TC) + c;
#pragma unroll
                    for (int e = 0; e < NM; ++e) mu[e] += other[e * TC];
#pragma unroll
                    for (int a = 0; a < NO; ++a) nu[a] += other[(NM + a) * TC];
                }
            }
        }
        if (wave == 0 && h == 0 && valid) {
#pragma unroll
            for (int e = 0; e < NM; ++e) ws[(long long)e * p.ncases + j] = mu[e];
#pragma unroll
            for (int a = 0; a < NO; ++a) ws[(long long)(NM + a) * p.ncases + j] = nu[a];
        }
        __syncthreads();
    }
}

// Kernel B: expand the normal equations from the moments, eliminate knowns, LDL^T, substitution.
template <int ORDER>
__global__ __launch_bounds__(64) void moment_solve_kernel(const KParams p, const double* __restrict__ ws) {
    constexpr int NO = ndofs(2, ORDER), NM = nmoments(ORDER), NE = NO * (NO + 1) / 2;
    const long long j = (long long)blockIdx.x * 64 + threadIdx.x;
    if (j >= p.ncases) return;
    unsigned long long known, dropped;
    effective_mask<NO>(p.knowns[j * p.sknowns], known, dropped);
    constexpr unsigned long long FULL = (1ull << NO) - 1ull;
    if (known == FULL) return;
    double M[NE], g[NO];
#pragma unroll
    for (int a = 0; a < NO; ++a) {
        constexpr int dummy = 0; (void)dummy;
        const int pa = Mono<2>::P[a], qa = Mono<2>::Q[a];
        g[a] = ws[(long long)(NM + a) * p.ncases + j] * (inv_fact(pa) * inv_fact(qa));
    }
    // load each moment once, scatter it (times the factorial constants) to every entry that uses it
    double mu[NM];
#pragma unroll
    for (int e = 0; e < NM; ++e) mu[e] = ws[(long long)e * p.ncases + j];
#pragma unroll
    for (int a = 0; a < NO; ++a)
#pragma unroll
        for (int b = a; b < NO; ++b) {
            const int pa = Mono<2>::P[a], qa = Mono<2>::Q[a], pb = Mono<2>::P[b], qb = Mono<2>::Q[b];
            M[tri<NO>(a, b)] = mu[midx(pa + pb, qa + qb)] * (inv_fact(pa) * inv_fact(qa) * inv_fact(pb) * inv_fact(qb));
        }
    double* fio = p.fi + j * p.sfi_j;
    if (known) {
        double val[NO];
#pragma unroll
        for (int a = 0; a < NO; ++a) val[a] = (((known & ~dropped) >> a) & 1ull) ? fio[a] : 0.0;
        eliminate_knowns<NO>(M, g, known, val);
    }
    ldlt_factor<NO>(M);
    ldlt_solve<NO>(M, g);
#pragma unroll
    for (int a = 0; a < NO; ++a)
        if (!((known >> a) & 1ull)) fio[a] = g[a];
}

// per-device workspace for the moments, grown on demand (not freed until process exit)
static std::mutex g_ws_mutex;
static DevBuf g_ws[16];

template <int ORDER, int K, int KSPLIT, int LPC, int UNR, int MINW>
static int launch_moment(const KParams& p, hipStream_t stream) {
    using G = MomentGeom<ORDER, K, KSPLIT, LPC>;
    int dev = 0;
    WLSQM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) { set_error("device ordinal out of range"); return WLSQM_EVALUE; }
    double* ws = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        const size_t need = (size_t)G::NACC * (size_t)p.ncases * sizeof(double);
        if (g_ws[dev].n < need) {
            WLSQM_HIP_CHECK(hipStreamSynchronize(stream));      // the old buffer may still be in use by earlier launches
            int rc = g_ws[dev].alloc(need);
            if (rc != WLSQM_OK) return rc;
        }
        ws = g_ws[dev].as<double>();
    }
    const long long ntiles = (p.ncases + G::TC - 1) / G::TC;
    static int per_cu = 0, cus = 0;
    auto kern = moment_tile_kernel<ORDER, K, KSPLIT, LPC, UNR, MINW>;
    if (!cus) {
        hipDeviceProp_t prop;
        WLSQM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        if (G::LDS_BYTES > 64 * 1024)
            WLSQM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        int occ = 0;
        WLSQM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, G::NT, G::LDS_BYTES));
        per_cu = occ > 0 ? occ : 1;
        cus = prop.multiProcessorCount;
    }
    long long grid = (long long)per_cu * cus;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NT), G::LDS_BYTES, stream, p, ntiles, ws);
    WLSQM_HIP_CHECK(hipGetLastError());
    const long long blocks = (p.ncases + 63) / 64;
    hipLaunchKernelGGL((moment_solve_kernel<ORDER>), dim3((unsigned)blocks), dim3(64), 0, stream, p, (const double*)ws);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("moment");
    return WLSQM_OK;
}

int launch_fit_moment(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    if (dimension != 2 || p.do_sens || p.iterative || p.case_index || p.hoods) return WLSQM_OK;
    if (p.sxk_k != 2 || p.sxk_j != max_nk * 2 || p.sfk_k != 1 || p.sfk_j != max_nk || p.sxi_j < 2) return WLSQM_OK;
    if ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(p.fk)) & 15u) return WLSQM_OK;
    const char* v = getenv("WLSQM_TILE_VARIANT");
    const int var = v ? atoi(v) : 0;
    if (order == 4 && max_nk == 64) {        // C3
        *handled = true;
        switch (var) {
            case 1: return launch_moment<4, 64, 4, 2, 2, 2>(p, stream);
            case 2: return launch_moment<4, 64, 4, 2, 8, 2>(p, stream);
            case 3: return launch_moment<4, 64, 2, 2, 4, 2>(p, stream);
            case 4: return launch_moment<4, 64, 4, 1, 4, 2>(p, stream);
            case 5: return launch_moment<4, 64, 4, 2, 4, 2>(p, stream);
            case 6: return launch_moment<4, 64, 2, 2, 8, 2>(p, stream);
            case 7: return launch_moment<4, 64, 2, 2, 2, 2>(p, stream);
            case 8: return launch_moment<4, 64, 2, 4, 4, 2>(p, stream);
            case 9: return launch_moment<4, 64, 1, 4, 4, 2>(p, stream);
            default: return launch_moment<4, 64, 2, 2, 4, 2>(p, stream);   // best of the round-1 A/B
        }
    }
    return WLSQM_OK;
}

}  // namespace wlsqm
