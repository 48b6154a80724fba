// fit_tile_even.hip — dense input, every even K <= 64 of the shapes the staged one-lane-per-case kernel (fit_stage.hip) does NOT take
// (round 5: the fixed-K instantiations of 2D order 3, 3D order 2 and 2D order 2 from 32 neighbours on — dead behind the staged kernel since
// round 4, reachable only through WLSQM_HIP_STAGE=0 — are retired; with that switch those shapes now run the runtime-K kernels of
// fit_tilek.hip, and fit_tile.hip keeps its curated instantiations — C2's and C5's round-3 kernels among them — as the A/B family)
// One of the per-family dispatch tables of the fixed-K tile kernels (wlsqm_tile.hpp; see fit_tile.hip).
#include "wlsqm_tile.hpp"

namespace wlsqm {

int launch_fit_tile_even(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    const bool gather = p.hoods != nullptr;
    if (gather) return WLSQM_OK;
    // Every even K up to 64 (dense input) outside the shapes curated one by one in fit_tile.hip.  Shapes by interleaved A/B
    // (tools/tune.py, 1M cases, grids of 16 workgroups per resident slot), ms per launch:
    //  * PAIR: one wave per 32-case tile, two lanes per case, moments + direct fk, shares padded to a multiple of 4 slots (KC;
    //    the extra slots are masked like those of a ragged case) — the small neighbourhoods: 2D order 2 up to K = 30 (K = 8 /
    //    16 / 24 / 30: 0.054 / 0.097 / 0.123 / 0.175 against 0.092 / 0.121 / 0.145 / 0.200 with four lanes per case), 3D order
    //    2 up to 24 (K = 16 / 24: 0.162 / 0.206 against 0.246 / 0.286), order 1 up to 20 (2D K = 8 / 12: 0.038 / 0.062 against
    //    0.053 / 0.082), 2D order 3 up to 38 (K = 16 / 24 / 32: 0.132 / 0.169 / 0.201 against 0.215 / 0.243 / 0.276);
    //    (one lane per case on a 64-case one-wave tile: within 10 % either way at K = 8-16, 2x slower at K = 24 — not instantiated);
    //  * PAD: one wave per 16-case tile, four lanes per case, or two waves x two lanes per case (3D order 2 from K = 50, order
    //    1), shares padded to a multiple of 8 slots — 2D order 2 at K = 36 / 44 / 52 / 60: 0.209 / 0.257 / 0.287 / 0.326 against
    //    0.324 / 0.368 / 0.397 / 0.437 on the runtime-K kernels, 3D order 2 at K = 36 / 48 / 56 / 64: 0.371 / 0.403 / 0.508 / 0.560
    //    against 0.609 / 0.692 / 0.823 / 0.977, 2D order 1 at K = 40: 0.197 against 0.284, 3D order 1 at K = 24 / 40: 0.165 /
    //    0.249 against 0.202 / 0.333, 2D order 3 at K = 48 / 64: 0.332 / 0.401 against 0.432 / 0.591;
    //  * HALF: two waves per 64-case tile, one lane per case (3D order 2 at K = 28: 0.261 against 0.449);
    //  * 1D: two waves per 64-case tile, moment form, fk staged (order 2 at K = 6 / 10 / 20: 0.025 / 0.037 / 0.075 against 0.054
    //    / 0.067 / 0.100, order 4 at K = 12 / 24: 0.050 / 0.102 against 0.096 / 0.144).
#define PAIR_CASE(D, O, KK, UU)                                                                                          \
    if (dimension == D && order == O && max_nk == KK) {                                                                 \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, 1, 2, UU, 2, false, true, true, false, (KK + 3) / 4 * 4>(p, stream);         \
    }
#define PAD_CASE(D, O, KK, KS, LL, UU)                                                                                   \
    if (dimension == D && order == O && max_nk == KK) {                                                                 \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, KS, LL, UU, 2, false, true, true, false, (KK + 7) / 8 * 8>(p, stream);       \
    }
#define HALF_CASE(D, O, KK)                                                                                               \
    if (dimension == D && order == O && max_nk == KK) {                                                                 \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, 2, 1, 2, 2, false, true, true>(p, stream);                                    \
    }
#define DENSE_CASE(D, O, KK, ...)                                                                                         \
    if (dimension == D && order == O && max_nk == KK) {                                                                 \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, __VA_ARGS__>(p, stream);                                                      \
    }
    if (dimension == 1) {
        DENSE_CASE(1, 1, 2, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 2, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 2, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 2, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 4, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 4, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 4, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 4, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 6, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 6, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 6, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 6, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 8, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 3, 8, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 4, 8, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 1, 10, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 2, 10, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 3, 10, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 4, 10, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 1, 12, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 2, 12, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 3, 12, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 4, 12, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 1, 14, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 2, 14, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 3, 14, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 4, 14, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 1, 16, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 16, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 16, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 18, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 18, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 18, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 18, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 20, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 20, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 20, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 20, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 22, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 22, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 22, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 22, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 24, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 24, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 24, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 24, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 26, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 26, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 26, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 26, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 28, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 28, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 28, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 28, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 30, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 30, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 30, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 30, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 32, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 32, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 32, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 32, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 34, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 34, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 34, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 34, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 36, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 36, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 36, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 36, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 38, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 38, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 38, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 38, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 40, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 40, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 40, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 40, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 42, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 42, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 42, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 42, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 44, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 44, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 44, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 44, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 46, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 46, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 46, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 46, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 48, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 48, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 48, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 48, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 50, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 50, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 50, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 50, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 52, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 52, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 52, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 52, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 54, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 54, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 54, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 54, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 56, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 56, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 56, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 56, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 58, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 58, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 58, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 58, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 60, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 60, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 60, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 60, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 62, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 62, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 62, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 62, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 1, 64, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, 64, 2, 1, 4, 2, false, false, true)
        DENSE_CASE(1, 3, 64, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, 64, 2, 1, 4, 2, false, false, true)
    } else {
        PAIR_CASE(2, 2, 8, 4) PAIR_CASE(2, 2, 10, 6) PAIR_CASE(2, 2, 12, 6)
        PAIR_CASE(2, 2, 14, 8) PAIR_CASE(2, 2, 16, 8) PAIR_CASE(2, 2, 18, 10)
        PAIR_CASE(2, 2, 20, 10) PAIR_CASE(2, 2, 22, 12) PAIR_CASE(2, 2, 24, 12)
        PAIR_CASE(2, 2, 26, 14) PAIR_CASE(2, 2, 28, 14) PAIR_CASE(2, 2, 30, 8)
          
          
         PAIR_CASE(2, 1, 2, 2) PAIR_CASE(3, 1, 2, 2)
        PAIR_CASE(2, 1, 4, 2) PAIR_CASE(3, 1, 4, 2) PAIR_CASE(2, 1, 6, 4)
        PAIR_CASE(3, 1, 6, 4) PAIR_CASE(2, 1, 8, 4) PAIR_CASE(3, 1, 8, 4)
        PAIR_CASE(2, 1, 10, 6) PAIR_CASE(3, 1, 10, 6) PAIR_CASE(2, 1, 12, 6)
        PAIR_CASE(3, 1, 12, 6) PAIR_CASE(2, 1, 14, 8) PAIR_CASE(3, 1, 14, 8)
        PAIR_CASE(3, 1, 16, 8) PAIR_CASE(2, 1, 18, 10) PAIR_CASE(3, 1, 18, 10)
        PAIR_CASE(2, 1, 20, 10) PAIR_CASE(3, 1, 20, 10) 
          
          
          
          
         
        
          
          
          
          
          
          
          
          
          
          
         PAD_CASE(2, 1, 22, 2, 2, 2) PAD_CASE(3, 1, 22, 2, 2, 2)
        PAD_CASE(2, 1, 24, 2, 2, 2) PAD_CASE(3, 1, 24, 2, 2, 2) PAD_CASE(2, 1, 26, 2, 2, 2)
        PAD_CASE(3, 1, 26, 2, 2, 2) PAD_CASE(2, 1, 28, 2, 2, 2) PAD_CASE(3, 1, 28, 2, 2, 2)
        PAD_CASE(2, 1, 30, 2, 2, 2) PAD_CASE(3, 1, 30, 2, 2, 2) PAD_CASE(3, 1, 32, 2, 2, 2)
        PAD_CASE(2, 1, 34, 2, 2, 2) PAD_CASE(3, 1, 34, 2, 2, 2) PAD_CASE(2, 1, 36, 2, 2, 2)
        PAD_CASE(3, 1, 36, 2, 2, 2) PAD_CASE(2, 1, 38, 2, 2, 2) PAD_CASE(3, 1, 38, 2, 2, 2)
        PAD_CASE(2, 1, 40, 2, 2, 2) PAD_CASE(3, 1, 40, 2, 2, 2) PAD_CASE(2, 1, 42, 2, 2, 2)
        PAD_CASE(3, 1, 42, 2, 2, 2) PAD_CASE(2, 1, 44, 2, 2, 2) PAD_CASE(3, 1, 44, 2, 2, 2)
        PAD_CASE(2, 1, 46, 2, 2, 2) PAD_CASE(3, 1, 46, 2, 2, 2) PAD_CASE(2, 1, 48, 2, 2, 2)
        PAD_CASE(3, 1, 48, 2, 2, 2) PAD_CASE(2, 1, 50, 2, 2, 2) PAD_CASE(3, 1, 50, 2, 2, 2)
        PAD_CASE(2, 1, 52, 2, 2, 2) PAD_CASE(3, 1, 52, 2, 2, 2) PAD_CASE(2, 1, 54, 2, 2, 2)
        PAD_CASE(3, 1, 54, 2, 2, 2) PAD_CASE(2, 1, 56, 2, 2, 2) PAD_CASE(3, 1, 56, 2, 2, 2)
        PAD_CASE(2, 1, 58, 2, 2, 2) PAD_CASE(3, 1, 58, 2, 2, 2) PAD_CASE(2, 1, 60, 2, 2, 2)
        PAD_CASE(3, 1, 60, 2, 2, 2) PAD_CASE(2, 1, 62, 2, 2, 2) PAD_CASE(3, 1, 62, 2, 2, 2)
        PAD_CASE(2, 1, 64, 2, 2, 2) PAD_CASE(3, 1, 64, 2, 2, 2) 
          
          
          
         
    }
#undef PAIR_CASE
#undef PAD_CASE
#undef HALF_CASE
#undef DENSE_CASE
    return WLSQM_OK;
}

}  // namespace wlsqm
