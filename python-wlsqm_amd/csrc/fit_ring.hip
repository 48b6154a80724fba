// fit_ring.hip — dispatch of the one-kernel ring fits (wlsqm_ring.hpp) for DENSE contiguous input; the index-based
// instantiations are compiled in fit_ring_gather.hip.
#include "wlsqm_ring.hpp"

namespace wlsqm {

// Dense contiguous 2D order-4 batches (the tile path's eligibility: fit_tile.hip tile_eligible).
bool tile_dense_eligible(int dimension, const KParams& p, long long max_nk);

int launch_fit_ring(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    const char* noring = getenv("WLSQM_HIP_DISABLE_RING");       // A/B against the two-kernel moment path
    if (noring && noring[0] == '1') return WLSQM_OK;
    if (p.hoods || p.do_sens || p.iterative || p.case_index) return WLSQM_OK;
    const char* v = getenv("WLSQM_TILE_VARIANT");                // tools/tune.py: A/B of the ring shapes against the tile kernels
    const int var = v ? atoi(v) : 0;
    // 3D order 2 with 40 neighbour slots (BASELINE configs[4]): the ring shape is 5 % ahead of the one-wave tile kernel (interleaved
    // A/B at 1M cases: 0.341 against 0.359 ms; unroll 5 instead of 10: 0.350; compiled for two waves per SIMD it spills 524 B per
    // lane: 1.01 ms).  WLSQM_TILE_VARIANT = 1 keeps the tile kernel (tools/tune.py).
    // 2D order 2 with 32 slots (the headline shape) ties: 0.1616 ms (two waves per SIMD, 242 registers, no spills) against 0.1604 ms
    // for the tile kernel of fit_tile.hip — two unrelated designs at the same 5.3 TB/s: that shape sits at what the memory system
    // gives this access mix; the tile kernel stays (WLSQM_TILE_VARIANT = 60 selects the ring for A/B).
    if (dimension == 3 && order == 2 && max_nk == 40 && var != 1 && tile_dense_eligible(dimension, p, max_nk)) {
        *handled = true;
        return launch_ring_impl<3, 2, 40, 10, 1>(p, stream);
    }
    if (dimension == 2 && order == 2 && max_nk == 32 && var == 60 && tile_dense_eligible(dimension, p, max_nk)) {
        *handled = true;
        return launch_ring_impl<2, 2, 32, 8, 2>(p, stream);
    }
    if (dimension != 2 || order != 4) return WLSQM_OK;
    if (!tile_dense_eligible(dimension, p, max_nk)) return WLSQM_OK;
    // Every even K from 26 to 72 (400k cases, ms per launch, two-kernel moment path -> this kernel): K = 26 / 32 / 40 / 48 / 56 / 64:
    // 0.220 / 0.192 / 0.281 / 0.238 / 0.323 / 0.260 -> 0.179 / 0.184 / 0.189 / 0.212 / 0.228 / 0.213.  Below 26 the two paths tie
    // (K = 16 / 24: 0.147 / 0.167 against 0.151 / 0.170); beyond 64 a row needs two DMA instructions and the shares are padded
    // (masked loop): K = 66 / 68 / 70 / 72: 0.314 / 0.306 / 0.307 / 0.342 against 0.308 / 0.374 / 0.384 / 0.356 — kept up to 72, where
    // four ring slots still fit a CU; from 74 on the ring takes 43-56 KB of LDS (three waves per CU): K = 80 / 100: 0.424 / 0.644
    // against 0.361 / 0.504 — those stay on the two-kernel path.
#define RING_CASE(KK) if (max_nk == KK) { *handled = true; return launch_ring_impl<2, 4, KK, 16, 1>(p, stream); }
    RING_CASE(26) RING_CASE(28) RING_CASE(30) RING_CASE(32) RING_CASE(34) RING_CASE(36) RING_CASE(38) RING_CASE(40) RING_CASE(42) RING_CASE(44)
    RING_CASE(46) RING_CASE(48) RING_CASE(50) RING_CASE(52) RING_CASE(54) RING_CASE(56) RING_CASE(58) RING_CASE(60) RING_CASE(62) RING_CASE(64)
    RING_CASE(66) RING_CASE(68) RING_CASE(70) RING_CASE(72)
#undef RING_CASE
    return WLSQM_OK;
}

}  // namespace wlsqm
