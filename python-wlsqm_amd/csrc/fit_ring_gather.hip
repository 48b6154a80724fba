// fit_ring_gather.hip — the one-kernel ring fit (wlsqm_ring.hpp) on INDEX-BASED input: 2D order 4, point table rows of 16 bytes
// gathered straight into the LDS ring by per-lane DMA addresses.  Replaces the two-kernel moment path (gathering tile pass ->
// 480 B/case workspace -> moment_solve_kernel) for the neighbour counts below; the reference's harness shape
// (examples/wlsqm_example.py:103-133: order 4, neighbour lists from a k-d tree) is this layout.
#include "wlsqm_ring.hpp"

namespace wlsqm {

bool tile_moments_supported(int dimension, int order, const KParams& p, long long max_nk);   // fit_tile.hip: layout + alignment of the index-based tables

int launch_fit_ring_gather(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    if (!p.hoods || dimension != 2 || order != 4) return WLSQM_OK;
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    const char* noring = getenv("WLSQM_HIP_DISABLE_RING");       // A/B against the two-kernel moment path
    if (noring && noring[0] == '1') return WLSQM_OK;
    if (p.do_sens || p.iterative || p.case_index) return WLSQM_OK;
    if (!tile_moments_supported(dimension, order, p, max_nk)) return WLSQM_OK;
    if (reinterpret_cast<uintptr_t>(p.F) & 7u) return WLSQM_OK;
    // up to 64 slots: one DMA instruction per row, and ring + index buffer stay under 40 KB (four workgroups per CU)
#define RING_CASE(KK) if (max_nk == KK) { *handled = true; return launch_ring_impl<2, 4, KK, 16, 1, true>(p, stream); }
    RING_CASE(26) RING_CASE(28) RING_CASE(30) RING_CASE(32) RING_CASE(34) RING_CASE(36) RING_CASE(38) RING_CASE(40) RING_CASE(42) RING_CASE(44)
    RING_CASE(46) RING_CASE(48) RING_CASE(50) RING_CASE(52) RING_CASE(54) RING_CASE(56) RING_CASE(58) RING_CASE(60) RING_CASE(62) RING_CASE(64)
#undef RING_CASE
    return WLSQM_OK;
}

}  // namespace wlsqm
