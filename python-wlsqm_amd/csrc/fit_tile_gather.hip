// fit_tile_gather.hip — index-based input, every even K <= 64
// One of the per-family dispatch tables of the fixed-K tile kernels (wlsqm_tile.hpp; see fit_tile.hip).
#include "wlsqm_tile.hpp"

namespace wlsqm {

int launch_fit_tile_gather(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    const bool gather = p.hoods != nullptr;
    if (!gather || max_nk > 64) return WLSQM_OK;
    // Index-based input, every even K up to 64 of 2D orders 1-3 and 3D orders 1-2 (the sizes curated one by one in fit_tile.hip
    // excepted): shares padded to a multiple of 4 slots, fk staged (F is gathered like S).  Shapes by A/B (tools/tune_cloud.py,
    // 1M cases, ms per launch):
    //  * the small neighbourhoods, one wave per 32-case tile with two lanes per case — 2D order 2 at K = 8 / 12 / 20 / 28: 0.056 /
    //    0.075 / 0.103 / 0.141 against 0.079 / 0.089 / 0.130 / 0.147 with four waves per 64-case tile; 3D order 2 at K = 16 / 24:
    //    0.147 / 0.201 against 0.224 / 0.268; 2D order 3 at K = 24: 0.153 against 0.234;
    //  * four waves per 64-case tile for the middle sizes (against the runtime-K one-wave kernel: 2D order 2 at K = 36 / 40: 0.228 /
    //    0.239 against 0.310 / 0.337; 3D order 2 at K = 36 / 56 / 64: 0.376 / 0.678 / 0.741 against 0.602 / 0.934 / 1.120; 2D order 1
    //    at K = 40: 0.179 against 0.297; 2D order 3 at K = 48: 0.333 against 0.484);
    //  * two waves x two lanes per case for the large 2D and 3D order-1 sizes and 3D order 2 around K = 44-48 (2D order 2 at K = 52 /
    //    60: 0.346 / 0.373 against 0.411 / 0.457; 3D order 2 at K = 48: 0.442 against 0.768; 3D order 1 at K = 40 / 56: 0.275 / 0.474
    //    against 0.359 / 0.552; 2D order 3 at K = 64: 0.445 against 0.584).
#define GATHER_CASE(D, O, KK, KS, LL, UU)                                                                                \
    if (dimension == D && order == O && max_nk == KK) {                                                                 \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, KS, LL, UU, 2, true, false, true, false, (KK + 3) / 4 * 4>(p, stream);       \
    }
    GATHER_CASE(2, 2, 8, 1, 2, 4) GATHER_CASE(2, 2, 10, 1, 2, 6) GATHER_CASE(2, 2, 12, 1, 2, 6)
    GATHER_CASE(2, 2, 14, 1, 2, 8) GATHER_CASE(2, 2, 16, 1, 2, 8) GATHER_CASE(2, 2, 18, 1, 2, 10)
    GATHER_CASE(2, 2, 20, 1, 2, 10) GATHER_CASE(2, 2, 22, 1, 2, 12) GATHER_CASE(2, 2, 24, 1, 2, 12)
    GATHER_CASE(2, 2, 26, 1, 2, 14) GATHER_CASE(2, 2, 28, 1, 2, 14) GATHER_CASE(2, 2, 30, 1, 2, 8)
    GATHER_CASE(2, 2, 34, 4, 1, 4) GATHER_CASE(2, 2, 36, 4, 1, 4) GATHER_CASE(2, 2, 38, 4, 1, 4)
    GATHER_CASE(2, 2, 40, 4, 1, 4) GATHER_CASE(2, 2, 42, 4, 1, 4) GATHER_CASE(2, 2, 44, 4, 1, 4)
    GATHER_CASE(2, 2, 46, 4, 1, 4) GATHER_CASE(2, 2, 50, 2, 2, 4) GATHER_CASE(2, 2, 52, 2, 2, 4)
    GATHER_CASE(2, 2, 54, 2, 2, 4) GATHER_CASE(2, 2, 56, 2, 2, 4) GATHER_CASE(2, 2, 58, 2, 2, 4)
    GATHER_CASE(2, 2, 60, 2, 2, 4) GATHER_CASE(2, 2, 62, 2, 2, 4) GATHER_CASE(3, 2, 12, 1, 2, 2)
    GATHER_CASE(3, 2, 14, 1, 2, 2) GATHER_CASE(3, 2, 16, 1, 2, 2) GATHER_CASE(3, 2, 18, 1, 2, 2)
    GATHER_CASE(3, 2, 20, 1, 2, 2) GATHER_CASE(3, 2, 22, 1, 2, 2) GATHER_CASE(3, 2, 24, 1, 2, 2)
    GATHER_CASE(3, 2, 26, 4, 1, 2) GATHER_CASE(3, 2, 28, 4, 1, 2) GATHER_CASE(3, 2, 30, 4, 1, 2)
    GATHER_CASE(3, 2, 34, 4, 1, 2) GATHER_CASE(3, 2, 36, 4, 1, 2) GATHER_CASE(3, 2, 38, 4, 1, 2)
    GATHER_CASE(3, 2, 42, 2, 2, 2) GATHER_CASE(3, 2, 44, 2, 2, 2) GATHER_CASE(3, 2, 46, 2, 2, 2)
    GATHER_CASE(3, 2, 48, 2, 2, 2) GATHER_CASE(3, 2, 50, 4, 1, 2) GATHER_CASE(3, 2, 52, 4, 1, 2)
    GATHER_CASE(3, 2, 54, 4, 1, 2) GATHER_CASE(3, 2, 56, 4, 1, 2) GATHER_CASE(3, 2, 58, 4, 1, 2)
    GATHER_CASE(3, 2, 60, 4, 1, 2) GATHER_CASE(3, 2, 62, 4, 1, 2) GATHER_CASE(3, 2, 64, 4, 1, 2)
    GATHER_CASE(2, 1, 4, 1, 2, 2) GATHER_CASE(2, 1, 6, 1, 2, 4) GATHER_CASE(2, 1, 8, 1, 2, 4)
    GATHER_CASE(2, 1, 10, 1, 2, 6) GATHER_CASE(2, 1, 12, 1, 2, 6) GATHER_CASE(2, 1, 14, 1, 2, 8)
    GATHER_CASE(2, 1, 18, 1, 2, 10) GATHER_CASE(2, 1, 20, 1, 2, 10) GATHER_CASE(2, 1, 22, 4, 1, 4)
    GATHER_CASE(2, 1, 24, 4, 1, 4) GATHER_CASE(2, 1, 26, 4, 1, 4) GATHER_CASE(2, 1, 28, 4, 1, 4)
    GATHER_CASE(2, 1, 30, 4, 1, 4) GATHER_CASE(2, 1, 34, 4, 1, 4) GATHER_CASE(2, 1, 36, 4, 1, 4)
    GATHER_CASE(2, 1, 38, 4, 1, 4) GATHER_CASE(2, 1, 40, 4, 1, 4) GATHER_CASE(2, 1, 42, 4, 1, 4)
    GATHER_CASE(2, 1, 44, 4, 1, 4) GATHER_CASE(2, 1, 46, 4, 1, 4) GATHER_CASE(2, 1, 48, 4, 1, 4)
    GATHER_CASE(2, 1, 50, 4, 1, 4) GATHER_CASE(2, 1, 52, 4, 1, 4) GATHER_CASE(2, 1, 54, 4, 1, 4)
    GATHER_CASE(2, 1, 56, 4, 1, 4) GATHER_CASE(2, 1, 58, 4, 1, 4) GATHER_CASE(2, 1, 60, 4, 1, 4)
    GATHER_CASE(2, 1, 62, 4, 1, 4) GATHER_CASE(2, 1, 64, 4, 1, 4) GATHER_CASE(3, 1, 4, 1, 2, 2)
    GATHER_CASE(3, 1, 6, 1, 2, 4) GATHER_CASE(3, 1, 8, 1, 2, 4) GATHER_CASE(3, 1, 10, 1, 2, 6)
    GATHER_CASE(3, 1, 12, 1, 2, 6) GATHER_CASE(3, 1, 14, 1, 2, 8) GATHER_CASE(3, 1, 16, 1, 2, 8)
    GATHER_CASE(3, 1, 18, 1, 2, 10) GATHER_CASE(3, 1, 20, 1, 2, 10) GATHER_CASE(3, 1, 22, 4, 1, 4)
    GATHER_CASE(3, 1, 24, 4, 1, 4) GATHER_CASE(3, 1, 26, 4, 1, 4) GATHER_CASE(3, 1, 28, 4, 1, 4)
    GATHER_CASE(3, 1, 30, 4, 1, 4) GATHER_CASE(3, 1, 34, 2, 2, 4) GATHER_CASE(3, 1, 36, 2, 2, 4)
    GATHER_CASE(3, 1, 38, 2, 2, 4) GATHER_CASE(3, 1, 40, 2, 2, 4) GATHER_CASE(3, 1, 42, 2, 2, 4)
    GATHER_CASE(3, 1, 44, 2, 2, 4) GATHER_CASE(3, 1, 46, 2, 2, 4) GATHER_CASE(3, 1, 48, 2, 2, 4)
    GATHER_CASE(3, 1, 50, 2, 2, 4) GATHER_CASE(3, 1, 52, 2, 2, 4) GATHER_CASE(3, 1, 54, 2, 2, 4)
    GATHER_CASE(3, 1, 56, 2, 2, 4) GATHER_CASE(3, 1, 58, 2, 2, 4) GATHER_CASE(3, 1, 60, 2, 2, 4)
    GATHER_CASE(3, 1, 62, 2, 2, 4) GATHER_CASE(3, 1, 64, 2, 2, 4) GATHER_CASE(2, 3, 12, 1, 2, 2)
    GATHER_CASE(2, 3, 14, 1, 2, 2) GATHER_CASE(2, 3, 16, 1, 2, 2) GATHER_CASE(2, 3, 18, 1, 2, 2)
    GATHER_CASE(2, 3, 20, 1, 2, 2) GATHER_CASE(2, 3, 22, 1, 2, 2) GATHER_CASE(2, 3, 24, 1, 2, 2)
    GATHER_CASE(2, 3, 26, 1, 2, 2) GATHER_CASE(2, 3, 28, 1, 2, 2) GATHER_CASE(2, 3, 30, 1, 2, 2)
    GATHER_CASE(2, 3, 32, 1, 2, 2) GATHER_CASE(2, 3, 34, 1, 2, 2) GATHER_CASE(2, 3, 36, 1, 2, 2)
    GATHER_CASE(2, 3, 38, 1, 2, 2) GATHER_CASE(2, 3, 42, 4, 1, 4) GATHER_CASE(2, 3, 44, 4, 1, 4)
    GATHER_CASE(2, 3, 46, 4, 1, 4) GATHER_CASE(2, 3, 48, 4, 1, 4) GATHER_CASE(2, 3, 50, 4, 1, 4)
    GATHER_CASE(2, 3, 52, 4, 1, 4) GATHER_CASE(2, 3, 54, 4, 1, 4) GATHER_CASE(2, 3, 56, 4, 1, 4)
    GATHER_CASE(2, 3, 58, 2, 2, 4) GATHER_CASE(2, 3, 60, 2, 2, 4) GATHER_CASE(2, 3, 62, 2, 2, 4)
    GATHER_CASE(2, 3, 64, 2, 2, 4)
#undef GATHER_CASE
    return WLSQM_OK;
}

}  // namespace wlsqm
