// fit_tile.hip — contiguous fast path of the WLSQM fit for gfx950: LDS-staged tiles.
//
// Same arithmetic as fit_lane.hip (one lane owns one local fit, normal matrix in VGPRs; see
// wlsqm_kernels.hpp for the reference citations), but the dense reference layout
// xk[ncases, K, dim], fk[ncases, K] is "array of structures" for a lane-per-case mapping, so
// the tile kernel moves it through LDS:
//
//   * a workgroup of 64*KSPLIT threads owns a tile of 64 consecutive cases; the tile's xk and
//     fk blocks are single contiguous byte ranges in HBM and are read with fully coalesced
//     16-byte-per-lane loads, ALL issued before the first is consumed (K*(dim+1)/(2*KSPLIT)
//     loads in flight per lane), then parked in LDS with a padded row per case;
//   * lane c of every wave reads row c back with conflict-free ds_read_b128/b64 (row stride
//     chosen so that 16/32 consecutive lanes cover all 64 banks);
//   * the K neighbours are split over the KSPLIT waves (k ascending inside each share); the
//     partial normal matrices are combined through LDS (reusing the tile's storage) and wave 0
//     does knowns elimination + LDL^T + substitution and writes the `no` results.
//
// LDS per workgroup is 64 * K * (dim+1) * 8 B plus padding (49.9 KB for 2D/32 neighbours), so
// three workgroups share a CU's 160 KB and 12 waves are resident with KSPLIT = 4; while one
// workgroup computes, the others have their ~48 KB of loads in flight, which is what keeps
// HBM busy (no intra-workgroup double buffering: LDS is the scarce resource here).
#include <cstdlib>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

namespace wlsqm {

constexpr int TILE = 64;          // cases per tile = lanes per wave

typedef double double2_ __attribute__((ext_vector_type(2)));   // 16-byte aligned pair -> dwordx4 / ds_*_b128

// round `v` up to the next value that is == r (mod m)
__host__ __device__ constexpr int round_up_mod(int v, int m, int r) { return v + ((r - v % m) % m + m) % m; }

// LDS row strides (in doubles).  xk row: K*DIM doubles, read by lane c at c*RS.
//   DIM == 2: ds_read_b128 of (x,y): RS == 2 (mod 4) makes 16 consecutive lanes hit 16 distinct 4-bank slots.
//   otherwise ds_read_b64: RS odd makes 32 consecutive lanes hit 32 distinct 2-bank slots.
template <int DIM> __host__ __device__ constexpr int row_stride_x(int K) {
    return DIM == 2 ? round_up_mod(K * DIM, 4, 2) : round_up_mod(K * DIM, 2, 1);
}
__host__ __device__ constexpr int row_stride_f(int K) { return round_up_mod(K, 2, 1); }

// UNR: unroll factor of the neighbour loops (KPW = fully unrolled); MINW: min waves per SIMD for the
// register allocator (__launch_bounds__ 2nd argument).
template <int DIM, int ORDER, int K, int KSPLIT, int UNR, int MINW, bool PREFETCH>
__global__ __launch_bounds__(TILE * KSPLIT, MINW) void fit_tile_kernel(const KParams p, const long long ntiles) {
    constexpr int NO = ndofs(DIM, ORDER);
    constexpr int NE = NO * (NO + 1) / 2;
    constexpr int NT = TILE * KSPLIT;                      // threads per workgroup
    constexpr int RS = row_stride_x<DIM>(K);
    constexpr int FS = row_stride_f(K);
    constexpr int XCH = TILE * K * DIM / 2;                // 16-byte chunks in the tile's xk block
    constexpr int FCH = TILE * K / 2;                      // ... and in its fk block
    static_assert((K * DIM) % 2 == 0 && K % 2 == 0, "rows must be multiples of 16 bytes");
    static_assert(XCH % NT == 0 && FCH % NT == 0, "chunks must divide evenly over the workgroup");
    constexpr int NX = XCH / NT, NF = FCH / NT;            // chunks per thread
    constexpr int CPRX = K * DIM / 2, CPRF = K / 2;        // chunks per row
    constexpr int KPW = K / KSPLIT;                        // neighbours per wave
    static_assert(K % KSPLIT == 0, "K must split evenly over the waves");
    constexpr int NRED = NE + NO + 1;                      // partials per lane: M, g, max_d2
    constexpr int LDS_TILE = TILE * (RS + FS);
    constexpr int LDS_RED = (KSPLIT - 1) * NRED * TILE;
    constexpr int LDS_DOUBLES = LDS_TILE > LDS_RED ? LDS_TILE : LDS_RED;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sX = lds;                                      // [TILE][RS]
    double* sF = lds + TILE * RS;                          // [TILE][FS]
    double* sMax = lds + LDS_DOUBLES;                      // [KSPLIT][TILE] partial max_d2 (outside the reused region)

    const int tid = threadIdx.x;
    const int lane = tid & (TILE - 1);
    const int wave = tid / TILE;                           // wave-uniform

    // Issue every global load of one tile (coalesced 16 B per lane); nothing waits on them here.
    double2_ bx[NX], bf[NF];
    auto issue_loads = [&](long long t) {
        const long long j0 = t * TILE;
        const long long nvalid = (p.ncases - j0 < TILE) ? (p.ncases - j0) : TILE;
        const double2_* gx = reinterpret_cast<const double2_*>(p.xk + j0 * (long long)(K * DIM));
        const double2_* gf = reinterpret_cast<const double2_*>(p.fk + j0 * (long long)K);
        const long long xlim = nvalid * CPRX, flim = nvalid * CPRF;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const long long q = tid + (long long)i * NT;
            bx[i] = gx[q < xlim ? q : xlim - 1];
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const long long q = tid + (long long)i * NT;
            bf[i] = gf[q < flim ? q : flim - 1];
        }
    };
    if constexpr (PREFETCH) { if ((long long)blockIdx.x < ntiles) issue_loads(blockIdx.x); }

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long j0 = tile * TILE;
        const long long j = j0 + lane;
        const bool valid = j < p.ncases;
        const long long jc = valid ? j : p.ncases - 1;     // clamp: tail lanes replay the last case, never store

        // ---- stage 1: the tile's loads (already in flight from the previous iteration when PREFETCH)
        if constexpr (!PREFETCH) issue_loads(tile);
        // per-case scalars (small, straight to registers)
        const int nkc = min(p.nk[jc * p.snk], K);
        const bool uniform = (p.wm[jc * p.swm] == WLSQM_WEIGHT_UNIFORM);
        unsigned long long known, dropped;
        effective_mask<NO>(p.knowns[jc * p.sknowns], known, dropped);
        double xi[DIM];
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.xi[jc * p.sxi_j + m];

        // ---- stage 2: park the tile in LDS (padded rows)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int q = tid + i * NT;
            const int r = q / CPRX, c2 = q - r * CPRX;     // compile-time divisors
            double* d = sX + r * RS + 2 * c2;
            if constexpr (RS % 2 == 0) *reinterpret_cast<double2_*>(d) = bx[i];
            else { d[0] = bx[i].x; d[1] = bx[i].y; }
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int q = tid + i * NT;
            const int r = q / CPRF, c2 = q - r * CPRF;
            double* d = sF + r * FS + 2 * c2;
            d[0] = bf[i].x; d[1] = bf[i].y;
        }
        __syncthreads();
        // software prefetch: the next tile's loads fly while this one is computed (costs NX+NF staging registers)
        if constexpr (PREFETCH) { if (tile + gridDim.x < ntiles) issue_loads(tile + gridDim.x); }

        const double* xr = sX + lane * RS;
        const double* fr = sF + lane * FS;
        const int k0 = wave * KPW;

        // Both passes run a fixed KPW iterations (fully unrolled, all ds_reads hoistable); neighbours
        // k >= nk[j] of a ragged case are neutralised by zero offset + zero weight instead of a branch.

        // ---- pass 1: largest squared distance of the case (impl.pyx:389-391): each wave scans its share,
        // the KSPLIT partial maxima meet in LDS
        double max_d2 = 0.0;
#pragma unroll UNR
        for (int kk = 0; kk < KPW; ++kk) {
            const int k = k0 + kk;
            double d2 = 0.0;
#pragma unroll
            for (int m = 0; m < DIM; ++m) { const double dd = xr[k * DIM + m] - xi[m]; d2 += dd * dd; }
            d2 = (k < nkc) ? d2 : 0.0;
            max_d2 = d2 > max_d2 ? d2 : max_d2;
        }
        if constexpr (KSPLIT > 1) {
            sMax[wave * TILE + lane] = max_d2;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < KSPLIT; ++w) { const double o = sMax[w * TILE + lane]; max_d2 = o > max_d2 ? o : max_d2; }
        }
        const double inv_max = inverse_max(max_d2);

        // ---- pass 2: this wave's share of the neighbours.  A wave whose 64 cases all use the full K
        // neighbours (the common case) runs the loop without the per-neighbour `live` selects.
        double M[NE], g[NO];
#pragma unroll
        for (int e = 0; e < NE; ++e) M[e] = 0.0;
#pragma unroll
        for (int a = 0; a < NO; ++a) g[a] = 0.0;
        auto neighbour = [&](int k, bool live) {
            double d[DIM], c[NO];
            if constexpr (DIM == 2) {
                const double2_ xy = *reinterpret_cast<const double2_*>(xr + 2 * k);   // ds_read_b128
                d[0] = xy.x - xi[0]; d[1] = xy.y - xi[1];
            } else {
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = xr[k * DIM + m] - xi[m];
            }
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = live ? d[m] : 0.0;
            const double d2 = monomials<DIM, ORDER>(d, c);
            const double w = live ? weight(d2, inv_max, uniform) : 0.0;
            const double f = live ? fr[k] : 0.0;
            accumulate<NO>(M, g, c, w, f);
        };
        if (__all(nkc >= K)) {
#pragma unroll UNR
            for (int kk = 0; kk < KPW; ++kk) neighbour(k0 + kk, true);
        } else {
#pragma unroll 1
            for (int kk = 0; kk < KPW; ++kk) neighbour(k0 + kk, k0 + kk < nkc);
        }

        // ---- combine the KSPLIT partial sums through LDS (the tile's storage is dead now)
        if constexpr (KSPLIT > 1) {
            __syncthreads();
            double* red = lds;
            if (wave > 0) {
                double* mine = red + (wave - 1) * (NRED * TILE) + lane;
#pragma unroll
                for (int e = 0; e < NE; ++e) mine[e * TILE] = M[e];
#pragma unroll
                for (int a = 0; a < NO; ++a) mine[(NE + a) * TILE] = g[a];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 1; w < KSPLIT; ++w) {
                    const double* other = red + (w - 1) * (NRED * TILE) + lane;
#pragma unroll
                    for (int e = 0; e < NE; ++e) M[e] += other[e * TILE];
#pragma unroll
                    for (int a = 0; a < NO; ++a) g[a] += other[(NE + a) * TILE];
                }
            }
        }

        // ---- wave 0: knowns elimination, LDL^T, substitution, store
        if (wave == 0) {
            constexpr unsigned long long FULL = (1ull << NO) - 1ull;
            if (valid && known != FULL) {
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
        }
        __syncthreads();   // the next tile overwrites LDS
    }
}

template <int DIM, int ORDER, int K, int KSPLIT, int UNR = 2, int MINW = 3, bool PREFETCH = false>
static int launch_tile(const KParams& p, hipStream_t stream) {
    constexpr int NO = ndofs(DIM, ORDER);
    constexpr int NE = NO * (NO + 1) / 2;
    constexpr int NRED = NE + NO + 1;
    constexpr int RS = row_stride_x<DIM>(K), FS = row_stride_f(K);
    constexpr int LDS_TILE = TILE * (RS + FS), LDS_RED = (KSPLIT - 1) * NRED * TILE;
    constexpr size_t lds_bytes = sizeof(double) * ((LDS_TILE > LDS_RED ? LDS_TILE : LDS_RED) + TILE * KSPLIT);
    static_assert(lds_bytes <= 160 * 1024, "tile does not fit LDS");
    const long long ntiles = (p.ncases + TILE - 1) / TILE;
    static int per_cu = 0, cus = 0;
    auto kern = fit_tile_kernel<DIM, ORDER, K, KSPLIT, UNR, MINW, PREFETCH>;
    if (!cus) {
        int dev = 0;
        WLSQM_HIP_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        WLSQM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        if (lds_bytes > 64 * 1024)
            WLSQM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        int occ = 0;
        WLSQM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, TILE * KSPLIT, lds_bytes));
        per_cu = occ > 0 ? occ : 1;
        cus = prop.multiProcessorCount;
    }
    long long grid = (long long)per_cu * cus;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(TILE * KSPLIT), lds_bytes, stream, p, ntiles);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("tile");
    return WLSQM_OK;
}

// The tile path needs: no extras, all cases in order, dense contiguous arrays, 16-byte aligned bases.
static bool tile_eligible(int dim, const KParams& p, long long K) {
    if (p.do_sens || p.iterative || p.case_index) return false;
    if (p.sxk_k != dim || p.sxk_j != K * dim || p.sfk_k != 1 || p.sfk_j != K) return false;
    if ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(p.fk)) & 15u) return false;
    return true;
}

int launch_fit_tile(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    // WLSQM_HIP_DISABLE_TILE=1 forces the generic kernels (A/B measurements and the tile-vs-lane parity test)
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    if (!tile_eligible(dimension, p, max_nk)) return WLSQM_OK;
#define TILE_CASE(D, O, KK, S)                                                   \
    if (dimension == D && order == O && max_nk == KK) { *handled = true; return launch_tile<D, O, KK, S>(p, stream); }
    if (dimension == 2 && order == 2 && max_nk == 32) {      // tuning variants, selected by WLSQM_TILE_VARIANT (default 0)
        const char* v = getenv("WLSQM_TILE_VARIANT");
        const int var = v ? atoi(v) : 0;
        *handled = true;
        switch (var) {
            case 1: return launch_tile<2, 2, 32, 4, 1, 3>(p, stream);
            case 2: return launch_tile<2, 2, 32, 4, 8, 2>(p, stream);
            case 4: return launch_tile<2, 2, 32, 4, 4, 3>(p, stream);
            case 5: return launch_tile<2, 2, 32, 2, 2, 2>(p, stream);
            case 6: return launch_tile<2, 2, 32, 2, 4, 2>(p, stream);
            case 7: return launch_tile<2, 2, 32, 4, 2, 2, true>(p, stream);
            case 8: return launch_tile<2, 2, 32, 4, 4, 2, true>(p, stream);
            case 9: return launch_tile<2, 2, 32, 4, 1, 2, true>(p, stream);
            case 10: return launch_tile<2, 2, 32, 4, 2, 3, true>(p, stream);
            case 11: return launch_tile<2, 2, 32, 4, 2, 3>(p, stream);
            default: return launch_tile<2, 2, 32, 4, 8, 2>(p, stream);   // best of the round-1 A/B (tools/tune.py)
        }
    }
    TILE_CASE(2, 2, 16, 4)
    TILE_CASE(2, 2, 24, 4)
    TILE_CASE(2, 2, 48, 4)
    TILE_CASE(2, 2, 64, 4)
    TILE_CASE(2, 1, 16, 4)
    TILE_CASE(2, 1, 32, 4)
    TILE_CASE(1, 2, 8, 2)
    TILE_CASE(1, 2, 16, 4)
    TILE_CASE(3, 1, 32, 4)
#undef TILE_CASE
    return WLSQM_OK;
}

}  // namespace wlsqm
