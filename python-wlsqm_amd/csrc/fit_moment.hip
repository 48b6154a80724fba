// fit_moment.hip — two-kernel moment path for the large 2D systems (order 4: 15 unknowns, C3).
//
// For order 4 the normal matrix has 120 unique entries plus 15 right-hand-side entries; keeping them in
// VGPRs next to the streaming pass costs ~270 registers per lane, and every single-kernel tile variant measured
// slower than the generic kernel.  But the 120 entries are only 45 distinct moments (wlsqm_moments.hpp), so:
//   A. fit_tile_kernel<..., MOM, SPLIT> (fit_tile.hip): the LDS-tiled streaming pass accumulates the 45 + 15
//      moments per case and parks them in a structure-of-arrays workspace, ws[e * stride + j] (480 B per case);
//   B. moment_solve_kernel (here): one lane per case expands M and g from the moments (compile-time factorial
//      constants), eliminates knowns, runs the in-register LDL^T and substitution and stores fi.
// The extra 960 B/case of workspace traffic is affordable because this configuration is fp64-VALU-bound.  Batches
// beyond 4M cases run in chunks, which bounds the workspace (stream-ordered allocation, hipMallocAsync) at 1.9 GB.  (Measured and rejected: chunks small enough
// to keep the workspace in the Infinity Cache, 32K-256K cases: 1M C3 cases took 1.10-0.77 ms instead of 0.73 ms —
// the launch boundaries and kernel tails cost more than the HBM round trip.)
#include <cstdlib>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_moments.hpp"

namespace wlsqm {

bool tile_moments_supported(int dimension, int order, const KParams& p, long long max_nk);
int launch_tile_moments(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled);

// Kernel B: expand the normal equations from the moments, eliminate knowns, LDL^T, substitution.  The 120 + 15 entries of
// an order-4 case take 256 VGPRs + 78 AGPRs: one wave per SIMD.  Capping the kernel at 256 registers for two waves per
// SIMD spills 324 B per lane and is slower (0.30 vs 0.17 ms per 1M cases).
template <int DIM, int ORDER>
__global__ __launch_bounds__(64) void moment_solve_kernel(const KParams p) {
    constexpr int NO = ndofs(DIM, ORDER), NM = mom_count<DIM>(2 * ORDER), NE = NO * (NO + 1) / 2;
    const long long j = (long long)blockIdx.x * 64 + threadIdx.x;
    if (j >= p.ncases) return;
    unsigned long long known, dropped;
    effective_mask<NO>(p.knowns[j * p.sknowns], known, dropped);
    constexpr unsigned long long FULL = (1ull << NO) - 1ull;
    if (known == FULL) return;
    const double* w = p.ws + j;
    // Every case of the wave has exactly F known (the reference's default knowns, BASELINE configs[2]): expand and factor the
    // reduced (NO - 1) system directly — 105 + 14 entries instead of 120 + 15 for 15 DOFs (see fit_ring.hip: bit-identical to the
    // generic path, whose first elimination step is the identity row).
    if constexpr (NO >= 3) {
        if (__all(known == 1ull && dropped == 0ull)) {
            constexpr int N1 = NO - 1, NE1 = N1 * (N1 + 1) / 2;
            double* fio1 = p.fi + j * p.sfi_j;
            const double v0 = fio1[0];
            double M1[NE1], r1[N1];
#pragma unroll
            for (int a = 1; a < NO; ++a) {
                const int pa = Mono<DIM>::P[a], qa = Mono<DIM>::Q[a], ra = Mono<DIM>::R[a];
                r1[a - 1] = w[(long long)(NM + mom_index<DIM>(pa, qa, ra)) * p.ws_stride] * (mom_inv_fact(pa) * mom_inv_fact(qa) * mom_inv_fact(ra));
            }
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const double m = w[(long long)i * p.ws_stride];
#pragma unroll
                for (int a = 1; a < NO; ++a) {
                    const int pa = Mono<DIM>::P[a], qa = Mono<DIM>::Q[a], ra = Mono<DIM>::R[a];
                    const double fa = mom_inv_fact(pa) * mom_inv_fact(qa) * mom_inv_fact(ra);
                    if (mom_index<DIM>(pa, qa, ra) == i) r1[a - 1] -= (m * (1.0 * fa)) * v0;
#pragma unroll
                    for (int b = a; b < NO; ++b) {
                        const int pb = Mono<DIM>::P[b], qb = Mono<DIM>::Q[b], rb = Mono<DIM>::R[b];
                        const double fb = mom_inv_fact(pb) * mom_inv_fact(qb) * mom_inv_fact(rb);
                        if (mom_index<DIM>(pa + pb, qa + qb, ra + rb) == i) M1[tri<N1>(a - 1, b - 1)] = m * (fa * fb);
                    }
                }
            }
            ldlt_factor<N1>(M1);
            ldlt_solve<N1>(M1, r1);
#pragma unroll
            for (int a = 1; a < NO; ++a) fio1[a] = r1[a - 1];
            return;
        }
    }
    double M[NE], g[NO];
    // every moment is loaded once (coalesced: consecutive lanes, consecutive cases) and scattered to its entries
    expand_moments_from<DIM, ORDER>([&](int i) { return w[(long long)i * p.ws_stride]; },
                                    [&](int i) { return w[(long long)(NM + i) * p.ws_stride]; }, M, g);
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

// cases per chunk
static long long chunk_cases() {
    const char* e = getenv("WLSQM_HIP_MOMENT_CHUNK");             // tuning override
    if (e) { const long long v = atoll(e); if (v > 0) return v; }
    return 4ll << 20;
}

template <int DIM, int ORDER>
static int launch_moment(const KParams& p0, long long max_nk, hipStream_t stream, bool* handled) {
    constexpr int NO = ndofs(DIM, ORDER), NACC = mom_count<DIM>(2 * ORDER) + NO;
    long long chunk = chunk_cases();
    if (chunk > p0.ncases) chunk = p0.ncases;
    // stream-ordered workspace from the library's private pool (wlsqm_internal.hpp): concurrent launches on other streams get
    // their own block, and the pool hands the same memory back to the next call on this stream without a device synchronisation
    double* ws = nullptr;
    {
        const int rc = scratch_alloc_async(reinterpret_cast<void**>(&ws), (size_t)NACC * (size_t)chunk * sizeof(double), stream);
        if (rc != WLSQM_OK) return rc;
    }
    for (long long j0 = 0; j0 < p0.ncases; j0 += chunk) {
        const long long n = (p0.ncases - j0 < chunk) ? (p0.ncases - j0) : chunk;
        KParams p = slice_cases(p0, j0, n);
        p.ws = ws; p.ws_stride = chunk;
        int rc = launch_tile_moments(DIM, ORDER, p, max_nk, stream, handled);
        if (rc != WLSQM_OK || !*handled) { (void)scratch_free_async(ws, stream); return rc; }
        const long long blocks = (n + 63) / 64;
        hipLaunchKernelGGL((moment_solve_kernel<DIM, ORDER>), dim3((unsigned)blocks), dim3(64), 0, stream, p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { (void)scratch_free_async(ws, stream); return hip_fail(e, "moment_solve_kernel"); }
    }
    { const int rc = scratch_free_async(ws, stream); if (rc != WLSQM_OK) return rc; }
    note_kernel(p0.hoods ? "moment-gather" : "moment");
    return WLSQM_OK;
}

int launch_fit_moment(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    if (p.do_sens || p.iterative || p.case_index) return WLSQM_OK;
    if (!tile_moments_supported(dimension, order, p, max_nk)) return WLSQM_OK;
    if (dimension == 2 && order == 4) return launch_moment<2, 4>(p, max_nk, stream, handled);
    return WLSQM_OK;
}

}  // namespace wlsqm
