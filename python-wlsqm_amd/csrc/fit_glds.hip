// fit_glds.hip — the one-wave tile fit with the xk tile brought in by LDS-DMA, one tile ahead (2D, K a power of two).
//
// fit_tile.hip's best dense shape (one wave per 16-case tile, 4 lanes per case, moment form, fk straight from global
// memory) still serialises, inside a wave, "issue the tile's loads -> wait -> park in LDS -> compute"; only the other
// waves of the CU hide the wait.  Here the wave's NEXT tile is already on its way while the current one is computed:
//   * xk: `global_load_lds_dwordx4` (__builtin_amdgcn_global_load_lds, 16 B per lane, no VGPR destination, no ds_write)
//     into the other half of a two-tile LDS ring.  The DMA writes LDS lane-linearly (1 KiB per wave-instruction), so
//     rows cannot be padded; the bank conflicts of 512-byte rows are removed on the SOURCE side instead: the lane that
//     fills chunk s of row c fetches neighbour (s - c) mod K, i.e. every row is stored rotated by its case number, and
//     lane (c, h) finds neighbour k at chunk (k + c) mod K — 16 consecutive lanes hit 16 different 4-bank groups.
//   * fk, nk, weighting, knowns, xi of the next tile: ordinary loads into a second register set.
// One `s_waitcnt vmcnt(0)` + barrier per tile (at the top, when the data had a whole tile's compute time to arrive).
#include <cstdlib>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_moments.hpp"

namespace wlsqm {

typedef double gd2_ __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int ORDER, int K, int UNR, int MINW>
__global__ __launch_bounds__(64, MINW) void fit_glds_kernel(const KParams p, const long long ntiles) {
    constexpr int DIM = 2, WV = 64, TC = 16, LPC = 4, KPL = K / LPC;
    constexpr int NO = ndofs(DIM, ORDER), NE = NO * (NO + 1) / 2, NM = mom_count<DIM>(2 * ORDER);
    constexpr int TILE = TC * K * DIM;                 // doubles per tile image
    constexpr int NI = TC * K / WV;                    // DMA instructions per tile (one 16-byte chunk = one neighbour)
    static_assert((K & (K - 1)) == 0 && K >= 16 && KPL % 2 == 0, "K must be a power of two >= 16");
    extern __shared__ __attribute__((aligned(16))) double lds[];          // [2][TILE]

    const int lane = threadIdx.x, c = lane % TC, h = lane / TC, k0 = h * KPL;

    struct Meta { int nk, wm; long long kn; double xi0, xi1; };
    Meta nxt;
    double fnext[KPL];

    // everything of tile `tile` that can be requested ahead of time: DMA of its xk rows into ring slot `slot`,
    // register loads of the per-case scalars and of this lane's fk values
    auto prefetch = [&](long long tile, int slot) {
        const long long j0 = tile * TC;
        const long long nvalid = (p.ncases - j0 < TC) ? (p.ncases - j0) : TC;
        const long long jc = (c < nvalid) ? j0 + c : p.ncases - 1;
        double* dst = lds + slot * TILE;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = i * WV + lane;               // LDS chunk this lane fills
            const int row = q / K, s = q % K;
            const long long r = (row < nvalid) ? row : nvalid - 1;        // tail tile: replay the last valid row
            const double* src = p.xk + ((j0 + r) * K + ((s - row) & (K - 1))) * DIM;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(dst + i * WV * DIM), 16, 0, 0);
        }
        nxt.nk = p.nk[jc * p.snk]; nxt.wm = p.wm[jc * p.swm]; nxt.kn = p.knowns[jc * p.sknowns];
        nxt.xi0 = p.xi[jc * p.sxi_j]; nxt.xi1 = p.xi[jc * p.sxi_j + 1];
        const gd2_* gr = reinterpret_cast<const gd2_*>(p.fk + jc * (long long)K + k0);
#pragma unroll
        for (int i = 0; i < KPL / 2; ++i) { const gd2_ v = gr[i]; fnext[2 * i] = v.x; fnext[2 * i + 1] = v.y; }
    };

    long long tile = blockIdx.x;
    if (tile < ntiles) prefetch(tile, 0);
    for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
        __syncthreads();                               // with a DMA in flight this is s_waitcnt vmcnt(0) + s_barrier
        const long long j = tile * TC + c;
        const bool valid = j < p.ncases;
        const int nkc = min(nxt.nk, K);
        const bool uniform = (nxt.wm == WLSQM_WEIGHT_UNIFORM);
        unsigned long long known, dropped;
        effective_mask<NO>(nxt.kn, known, dropped);
        const double xi[DIM] = {nxt.xi0, nxt.xi1};
        double f[KPL];
#pragma unroll
        for (int i = 0; i < KPL; ++i) f[i] = fnext[i];
        const double* row = lds + (it & 1) * TILE + c * (K * DIM);
        const long long nexttile = tile + gridDim.x;
        if (nexttile < ntiles) prefetch(nexttile, (it + 1) & 1);

        auto offset = [&](int k, double (&d)[DIM]) {   // neighbour k of this lane's case: chunk (k + c) mod K of its row
            const gd2_ xy = *reinterpret_cast<const gd2_*>(row + ((k + c) & (K - 1)) * DIM);
            d[0] = xy.x - xi[0]; d[1] = xy.y - xi[1];
        };
        double max_d2 = 0.0;
#pragma unroll UNR
        for (int kk = 0; kk < KPL; ++kk) {
            double d[DIM];
            offset(k0 + kk, d);
            double d2 = d[0] * d[0] + d[1] * d[1];
            d2 = (k0 + kk < nkc) ? d2 : 0.0;
            max_d2 = d2 > max_d2 ? d2 : max_d2;
        }
#pragma unroll
        for (int off = TC; off < WV; off <<= 1) { const double o = __shfl_xor(max_d2, off, WV); max_d2 = o > max_d2 ? o : max_d2; }
        const double inv_max = inverse_max(max_d2);

        double mu[NM], nu[NO];
#pragma unroll
        for (int e = 0; e < NM; ++e) mu[e] = 0.0;
#pragma unroll
        for (int a = 0; a < NO; ++a) nu[a] = 0.0;
        auto neighbour = [&](int kk, bool live) {
            double d[DIM];
            offset(k0 + kk, d);
            d[0] = live ? d[0] : 0.0; d[1] = live ? d[1] : 0.0;
            const double d2 = d[0] * d[0] + d[1] * d[1];
            const double w = live ? weight(d2, inv_max, uniform) : 0.0;
            accumulate_moments<DIM, ORDER>(mu, nu, d, w, live ? f[kk] : 0.0);
        };
        if (__all(nkc >= K)) {
#pragma unroll UNR
            for (int kk = 0; kk < KPL; ++kk) neighbour(kk, true);
        } else {
#pragma unroll 1
            for (int kk = 0; kk < KPL; ++kk) neighbour(kk, k0 + kk < nkc);
        }
#pragma unroll
        for (int off = TC; off < WV; off <<= 1) {
#pragma unroll
            for (int e = 0; e < NM; ++e) mu[e] += __shfl_xor(mu[e], off, WV);
#pragma unroll
            for (int a = 0; a < NO; ++a) nu[a] += __shfl_xor(nu[a], off, WV);
        }
        constexpr unsigned long long FULL = (1ull << NO) - 1ull;
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

template <int ORDER, int K, int UNR, int MINW>
static int launch_glds_impl(const KParams& p, hipStream_t stream) {
    constexpr size_t lds_bytes = sizeof(double) * 2 * 16 * K * 2;
    const long long ntiles = (p.ncases + 15) / 16;
    static int per_cu = 0, cus = 0;
    auto kern = fit_glds_kernel<ORDER, K, UNR, MINW>;
    if (!cus) {
        int dev = 0;
        WLSQM_HIP_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        WLSQM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        int occ = 0;
        WLSQM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, lds_bytes));
        per_cu = occ > 0 ? occ : 1;
        cus = prop.multiProcessorCount;
    }
    long long grid = (long long)per_cu * cus;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), lds_bytes, stream, p, ntiles);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("tile-glds");
    return WLSQM_OK;
}

// Dense contiguous 2D order-2 batches with 32 neighbour slots (C2).  `variant` = WLSQM_TILE_VARIANT.
int launch_fit_glds(int dimension, int order, const KParams& p, long long max_nk, int variant, hipStream_t stream, bool* handled) {
    *handled = false;
    if (dimension != 2 || order != 2 || max_nk != 32) return WLSQM_OK;
    *handled = true;
    switch (variant) {
        case 41: return launch_glds_impl<2, 32, 4, 2>(p, stream);
        case 42: return launch_glds_impl<2, 32, 8, 3>(p, stream);
        case 43: return launch_glds_impl<2, 32, 4, 3>(p, stream);
        default: return launch_glds_impl<2, 32, 8, 2>(p, stream);
    }
}

}  // namespace wlsqm
