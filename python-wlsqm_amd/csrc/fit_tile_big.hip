// fit_tile_big.hip — 64 < K <= 128, dense (order 1: the shapes of order 2-3 run the staged kernel since round 4; their dense
// instantiations here were retired in round 5) and index-based
// One of the per-family dispatch tables of the fixed-K tile kernels (wlsqm_tile.hpp; see fit_tile.hip).
#include "wlsqm_tile.hpp"

namespace wlsqm {

int launch_fit_tile_big(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    const bool gather = p.hoods != nullptr;
    if (max_nk <= 64 || max_nk > 128) return WLSQM_OK;
    // Large neighbourhoods (64 < K <= 128, e.g. the 124 neighbours of a 5 x 5 x 5 block): two waves x four lanes per case on
    // a 16-case tile, shares padded to a multiple of 16 slots.  Before, these sizes ran on the generic lane-per-case kernel:
    // 400k cases, 2D order 2 at K = 80 / 128: 0.206 / 0.288 ms against 1.13 / 1.81; 3D order 2 at K = 80 / 124: 0.321 / 0.599
    // against 0.950 / 1.684 (one wave per tile or two lanes per case: 0.26-0.59 / 0.45-1.86).
#define BIG_CASE(D, O, KK, UU)                                                                                            \
    if (!gather && dimension == D && order == O && max_nk == KK) {                                                      \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, 2, 4, UU, 2, false, true, true, false, (KK + 15) / 16 * 16>(p, stream);      \
    }
#define BIG_K(X, ...) X(__VA_ARGS__, 66) X(__VA_ARGS__, 68) X(__VA_ARGS__, 70) X(__VA_ARGS__, 72) X(__VA_ARGS__, 74) X(__VA_ARGS__, 76) \
    X(__VA_ARGS__, 78) X(__VA_ARGS__, 80) X(__VA_ARGS__, 82) X(__VA_ARGS__, 84) X(__VA_ARGS__, 86) X(__VA_ARGS__, 88) X(__VA_ARGS__, 90) \
    X(__VA_ARGS__, 92) X(__VA_ARGS__, 94) X(__VA_ARGS__, 96) X(__VA_ARGS__, 98) X(__VA_ARGS__, 100) X(__VA_ARGS__, 102) X(__VA_ARGS__, 104) \
    X(__VA_ARGS__, 106) X(__VA_ARGS__, 108) X(__VA_ARGS__, 110) X(__VA_ARGS__, 112) X(__VA_ARGS__, 114) X(__VA_ARGS__, 116) \
    X(__VA_ARGS__, 118) X(__VA_ARGS__, 120) X(__VA_ARGS__, 122) X(__VA_ARGS__, 124) X(__VA_ARGS__, 126) X(__VA_ARGS__, 128)
#define BIG_2D2(KK) BIG_CASE(2, 2, KK, 4)
#define BIG_3D2(KK) BIG_CASE(3, 2, KK, 2)
#define BIG_2D1(KK) BIG_CASE(2, 1, KK, 4)
#define BIG_3D1(KK) BIG_CASE(3, 1, KK, 4)
#define BIG_2D3(KK) BIG_CASE(2, 3, KK, 2)
#define BIG_ONE(F, KK) F(KK)
    if (!gather) {
        if (dimension == 2 && order == 1) { BIG_K(BIG_ONE, BIG_2D1) }
        if (dimension == 3 && order == 1) { BIG_K(BIG_ONE, BIG_3D1) }
    }
    // index-based input at these sizes (order 2): the same shape without direct fk, shares padded to a multiple of 8 slots;
    // 400k cases, 2D K = 80 / 128: 0.250 / 0.275 ms against 1.45 / 2.53 on the generic kernel, 3D K = 80 / 124: 0.355 / 0.699
    // against 0.944 / 1.440
#define BIGG_CASE(D, O, KK, UU)                                                                                            \
    if (gather && dimension == D && order == O && max_nk == KK) {                                                       \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, 2, 4, UU, 2, true, false, true, false, (KK + 7) / 8 * 8>(p, stream);         \
    }
#define BIGG_2D2(KK) BIGG_CASE(2, 2, KK, 4)
#define BIGG_3D2(KK) BIGG_CASE(3, 2, KK, 2)
#define BIGG_2D1(KK) BIGG_CASE(2, 1, KK, 4)
#define BIGG_3D1(KK) BIGG_CASE(3, 1, KK, 4)
#define BIGG_2D3(KK) BIGG_CASE(2, 3, KK, 2)
    if (gather) {
        if (dimension == 2 && order == 2) { BIG_K(BIG_ONE, BIGG_2D2) }
        if (dimension == 3 && order == 2) { BIG_K(BIG_ONE, BIGG_3D2) }
        if (dimension == 2 && order == 1) { BIG_K(BIG_ONE, BIGG_2D1) }
        if (dimension == 3 && order == 1) { BIG_K(BIG_ONE, BIGG_3D1) }
        if (dimension == 2 && order == 3) { BIG_K(BIG_ONE, BIGG_2D3) }
    }
#undef BIGG_2D2
#undef BIGG_3D2
#undef BIGG_2D1
#undef BIGG_3D1
#undef BIGG_2D3
#undef BIGG_CASE
#undef BIG_ONE
#undef BIG_2D2
#undef BIG_3D2
#undef BIG_2D1
#undef BIG_3D1
#undef BIG_2D3
#undef BIG_K
#undef BIG_CASE
    return WLSQM_OK;
}

}  // namespace wlsqm
