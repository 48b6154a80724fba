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
// INV: besides fi, the wave leaves the INVERSE of every case's eliminated normal matrix for the sensitivities (fit_sens.hip), `no`
// substitutions with unit vectors on the factor that is in registers anyway.  Layout inv[group of 64 cases][column][case][row]:
// column c of the wave's 64 cases is one contiguous run of 64 no doubles, staged through LDS and stored as full lines (rows and
// columns of known DOFs are zero).  No lane leaves early in this variant (the copy-out is cooperative): lanes past the end of
// the batch replay its last case and store nothing but their slot of the scratch block.
template <int DIM, int ORDER, bool INV = false>
__global__ __launch_bounds__(64) void moment_solve_kernel(const KParams p, double* __restrict__ inv) {
    constexpr int NO = ndofs(DIM, ORDER), NM = mom_count<DIM>(2 * ORDER), NE = NO * (NO + 1) / 2;
    __shared__ __attribute__((aligned(16))) double sL[64 * NO];
    const int lane = threadIdx.x;
    const long long j_raw = (long long)blockIdx.x * 64 + lane;
    const bool active = j_raw < p.ncases;
    const long long j = active ? j_raw : p.ncases - 1;
    unsigned long long known, dropped;
    effective_mask<NO>(p.knowns[j * p.sknowns], known, dropped);
    constexpr unsigned long long FULL = (1ull << NO) - 1ull;
    // The wave's 64 fi rows as ONE run of 16-byte pieces through LDS (a known DOF re-written with its own bits, infra.pyx:780-795)
    // when every lane has a case with unknowns and no dropped DOF, and the rows are contiguous and aligned; else 8-byte stores per
    // lane — the slow pattern of this memory system (csrc/fit_sens.hip).  No lane leaves early: the copy-out is cooperative.
    const bool run_store = p.sfi_j == NO && ((reinterpret_cast<uintptr_t>(p.fi) & 15u) == 0) &&
                           __all(active && dropped == 0ull && known != FULL);
    auto store_rows = [&](const double (&row)[NO]) {              // row[a]: the full fi row of this lane's case
        if (run_store) {
#pragma unroll
            for (int a = 0; a < NO; ++a) sL[lane * NO + a] = row[a];
            __syncthreads();
            typedef double md2_ __attribute__((ext_vector_type(2)));
            md2_* out = reinterpret_cast<md2_*>(p.fi + (long long)blockIdx.x * 64 * NO);
            const md2_* src = reinterpret_cast<const md2_*>(sL);
#pragma unroll
            for (int q = lane; q < 64 * NO / 2; q += 64) out[q] = src[q];
            __syncthreads();
        } else if (active && known != FULL) {
            double* fio = p.fi + j * p.sfi_j;
#pragma unroll
            for (int a = 0; a < NO; ++a)
                if (!((known >> a) & 1ull)) fio[a] = row[a];
        }
    };
    const double* w = p.ws + j;
    // column(col, o): column `col` of the inverse into o[0..NO)
    auto emit_inverse = [&](auto&& column) {
        double* blk = inv + (long long)blockIdx.x * (64 * NO * NO);
#pragma unroll 1
        for (int col = 0; col < NO; ++col) {
            double o[NO];
            column(col, o);
#pragma unroll
            for (int a = 0; a < NO; ++a) sL[lane * NO + a] = o[a];
            __syncthreads();
            double* dst = blk + col * (64 * NO);
#pragma unroll
            for (int i = 0; i < NO; ++i) dst[lane + 64 * i] = sL[lane + 64 * i];
            __syncthreads();
        }
    };
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
            {
                double row[NO];
                row[0] = v0;
#pragma unroll
                for (int a = 1; a < NO; ++a) row[a] = r1[a - 1];
                store_rows(row);
            }
            if constexpr (INV) {
                emit_inverse([&](int col, double (&o)[NO]) {
                    double sv[N1];
#pragma unroll
                    for (int a = 0; a < N1; ++a) sv[a] = (a + 1 == col) ? 1.0 : 0.0;      // column 0 (the known DOF): zeros
                    ldlt_solve<N1>(M1, sv);
                    o[0] = 0.0;
#pragma unroll
                    for (int a = 0; a < N1; ++a) o[a + 1] = sv[a];
                });
            }
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
    {
        double row[NO];
#pragma unroll
        for (int a = 0; a < NO; ++a) row[a] = ((known >> a) & 1ull) ? fio[a] : g[a];
        store_rows(row);
    }
    if constexpr (INV) {
        emit_inverse([&](int col, double (&o)[NO]) {
            const bool kcol = (known >> col) & 1ull;                     // known: zero row and column (the rows are identity rows)
#pragma unroll
            for (int a = 0; a < NO; ++a) o[a] = (a == col && !kcol) ? 1.0 : 0.0;
            ldlt_solve<NO>(M, o);
        });
    }
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
        hipLaunchKernelGGL((moment_solve_kernel<DIM, ORDER>), dim3((unsigned)blocks), dim3(64), 0, stream, p, (double*)nullptr);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { (void)scratch_free_async(ws, stream); return hip_fail(e, "moment_solve_kernel"); }
    }
    { const int rc = scratch_free_async(ws, stream); if (rc != WLSQM_OK) return rc; }
    note_kernel(p0.hoods ? "moment-gather" : "moment");
    return WLSQM_OK;
}

// The fit of ONE slice (p.ncases cases) that also leaves the inverses at inv[ceil(ncases / 64)][no][64][no] (first kernels of
// fit_sens.hip for 2D order 4).
bool moment_inverse_ok(int dimension, int order, const KParams& p, long long max_nk) {
    KParams q = p;
    q.do_sens = 0; q.sens = nullptr; q.iterative = 0;                    // (what the first kernel is launched with)
    return !p.case_index && tile_moments_supported(dimension, order, q, max_nk);
}
int launch_fit_moment_inverse(int dimension, int order, const KParams& p0, long long max_nk, double* inv, hipStream_t stream) {
    if (dimension != 2 || order != 4) { set_error("fit_moment_inverse: unsupported (dimension, order)"); return WLSQM_EVALUE; }
    constexpr int NACC = mom_count<2>(8) + ndofs(2, 4);
    double* ws = nullptr;
    int rc = scratch_alloc_async(reinterpret_cast<void**>(&ws), (size_t)NACC * (size_t)p0.ncases * sizeof(double), stream);
    if (rc != WLSQM_OK) return rc;
    KParams p = p0;
    p.ws = ws; p.ws_stride = p0.ncases; p.do_sens = 0; p.sens = nullptr; p.iterative = 0;
    bool handled = false;
    rc = launch_tile_moments(dimension, order, p, max_nk, stream, &handled);
    if (rc == WLSQM_OK && !handled) { set_error("fit_moment_inverse: no moment kernel for this shape"); rc = WLSQM_EVALUE; }
    if (rc == WLSQM_OK) {
        const long long blocks = (p.ncases + 63) / 64;
        hipLaunchKernelGGL((moment_solve_kernel<2, 4, true>), dim3((unsigned)blocks), dim3(64), 0, stream, p, inv);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = hip_fail(e, "moment_solve_kernel (inverse)");
    }
    const int rc2 = scratch_free_async(ws, stream);
    if (rc == WLSQM_OK) note_kernel("moment-inverse");
    return rc != WLSQM_OK ? rc : rc2;
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
