// fit_tile_gather.hip — index-based input, every even K <= 64
// One of the per-family dispatch tables of the fixed-K tile kernels (wlsqm_tile.hpp; see fit_tile.hip).
#include "wlsqm_tile.hpp"

namespace wlsqm {

int launch_fit_tile_gather(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    const bool gather = p.hoods != nullptr;
    if (!gather || max_nk > 64) return WLSQM_OK;
    // Index-based input (every even K up to 64, order 2): four waves per 64-case tile (two waves x two lanes per case for
    // the large 2D and the middle 3D sizes), shares padded to a multiple of 4 slots.  tools/tune_cloud.py, 1M cases, ms per
    // launch against the runtime-K one-wave kernel: 2D K = 20 / 36 / 40 / 52 / 60: 0.139 / 0.228 / 0.239 / 0.346 / 0.373
    // against 0.207 / 0.310 / 0.337 / 0.411 / 0.457; 3D K = 28 / 36 / 48 / 56 / 64: 0.309 / 0.376 / 0.442 / 0.678 / 0.741
    // against 0.382 / 0.602 / 0.768 / 0.934 / 1.120.
#define GATHER_CASE(D, O, KK, KS, LL, UU)                                                                                \
    if (gather && dimension == D && order == O && max_nk == KK) {                                                       \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, KS, LL, UU, 2, true, false, true, false, (KK + 3) / 4 * 4>(p, stream);       \
    }
    GATHER_CASE(2, 2, 8, 4, 1, 4) GATHER_CASE(2, 2, 10, 4, 1, 4) GATHER_CASE(2, 2, 12, 4, 1, 4)
    GATHER_CASE(2, 2, 14, 4, 1, 4) GATHER_CASE(2, 2, 18, 4, 1, 4) GATHER_CASE(2, 2, 20, 4, 1, 4)
    GATHER_CASE(2, 2, 22, 4, 1, 4) GATHER_CASE(2, 2, 26, 4, 1, 4) GATHER_CASE(2, 2, 28, 4, 1, 4)
    GATHER_CASE(2, 2, 30, 4, 1, 4) GATHER_CASE(2, 2, 34, 4, 1, 4) GATHER_CASE(2, 2, 36, 4, 1, 4)
    GATHER_CASE(2, 2, 38, 4, 1, 4) GATHER_CASE(2, 2, 40, 4, 1, 4) GATHER_CASE(2, 2, 42, 4, 1, 4)
    GATHER_CASE(2, 2, 44, 4, 1, 4) GATHER_CASE(2, 2, 46, 4, 1, 4) GATHER_CASE(2, 2, 50, 2, 2, 4)
    GATHER_CASE(2, 2, 52, 2, 2, 4) GATHER_CASE(2, 2, 54, 2, 2, 4) GATHER_CASE(2, 2, 56, 2, 2, 4)
    GATHER_CASE(2, 2, 58, 2, 2, 4) GATHER_CASE(2, 2, 60, 2, 2, 4) GATHER_CASE(2, 2, 62, 2, 2, 4)
    GATHER_CASE(3, 2, 12, 4, 1, 2) GATHER_CASE(3, 2, 14, 4, 1, 2) GATHER_CASE(3, 2, 16, 4, 1, 2)
    GATHER_CASE(3, 2, 18, 4, 1, 2) GATHER_CASE(3, 2, 20, 4, 1, 2) GATHER_CASE(3, 2, 22, 4, 1, 2)
    GATHER_CASE(3, 2, 24, 4, 1, 2) GATHER_CASE(3, 2, 26, 4, 1, 2) GATHER_CASE(3, 2, 28, 4, 1, 2)
    GATHER_CASE(3, 2, 30, 4, 1, 2) GATHER_CASE(3, 2, 34, 4, 1, 2) GATHER_CASE(3, 2, 36, 4, 1, 2)
    GATHER_CASE(3, 2, 38, 4, 1, 2) GATHER_CASE(3, 2, 42, 2, 2, 2) GATHER_CASE(3, 2, 44, 2, 2, 2)
    GATHER_CASE(3, 2, 46, 2, 2, 2) GATHER_CASE(3, 2, 48, 2, 2, 2) GATHER_CASE(3, 2, 50, 4, 1, 2)
    GATHER_CASE(3, 2, 52, 4, 1, 2) GATHER_CASE(3, 2, 54, 4, 1, 2) GATHER_CASE(3, 2, 56, 4, 1, 2)
    GATHER_CASE(3, 2, 58, 4, 1, 2) GATHER_CASE(3, 2, 60, 4, 1, 2) GATHER_CASE(3, 2, 62, 4, 1, 2)
    GATHER_CASE(3, 2, 64, 4, 1, 2)
#define EVEN_K4(X, ...) X(__VA_ARGS__, 4) X(__VA_ARGS__, 6) X(__VA_ARGS__, 8) X(__VA_ARGS__, 10) X(__VA_ARGS__, 12) X(__VA_ARGS__, 14) \
    X(__VA_ARGS__, 16) X(__VA_ARGS__, 18) X(__VA_ARGS__, 20) X(__VA_ARGS__, 22) X(__VA_ARGS__, 24) X(__VA_ARGS__, 26) X(__VA_ARGS__, 28) \
    X(__VA_ARGS__, 30) X(__VA_ARGS__, 32) X(__VA_ARGS__, 34) X(__VA_ARGS__, 36) X(__VA_ARGS__, 38) X(__VA_ARGS__, 40) X(__VA_ARGS__, 42) \
    X(__VA_ARGS__, 44) X(__VA_ARGS__, 46) X(__VA_ARGS__, 48) X(__VA_ARGS__, 50) X(__VA_ARGS__, 52) X(__VA_ARGS__, 54) X(__VA_ARGS__, 56) \
    X(__VA_ARGS__, 58) X(__VA_ARGS__, 60) X(__VA_ARGS__, 62) X(__VA_ARGS__, 64)
    // (the other families, same rule: four waves per 64-case tile, two waves x two lanes per case for the large sizes;
    // 2D order 1 at K = 10 / 20 / 40: 0.061 / 0.093 / 0.179 against 0.099 / 0.132 / 0.297 ms, 2D order 3 at K = 24 / 48: 0.242 /
    // 0.333 against 0.309 / 0.484, 3D order 1 at K = 16 / 24: 0.094 / 0.145 against 0.143 / 0.191)
#define GATHER_41(D, O, KK) GATHER_CASE(D, O, KK, 4, 1, 4)
#define GATHER_22(D, O, KK) GATHER_CASE(D, O, KK, 2, 2, 4)
    {
        if (dimension == 2 && order == 1 && max_nk != 16 && max_nk != 32) { EVEN_K4(GATHER_41, 2, 1) }
        if (dimension == 2 && order == 3 && max_nk >= 12 && max_nk <= 56 && max_nk != 40) { EVEN_K4(GATHER_41, 2, 3) }
        if (dimension == 2 && order == 3 && max_nk > 56) { GATHER_22(2, 3, 58) GATHER_22(2, 3, 60) GATHER_22(2, 3, 62) GATHER_22(2, 3, 64) }
        if (dimension == 3 && order == 1 && max_nk < 32) { EVEN_K4(GATHER_41, 3, 1) }
        if (dimension == 3 && order == 1 && max_nk > 32) { EVEN_K4(GATHER_22, 3, 1) }
    }
#undef GATHER_41
#undef GATHER_22
#undef EVEN_K4
#undef GATHER_CASE
    return WLSQM_OK;
}

}  // namespace wlsqm
