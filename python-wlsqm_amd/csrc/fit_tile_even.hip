// fit_tile_even.hip — dense input, every even K <= 64
// One of the per-family dispatch tables of the fixed-K tile kernels (wlsqm_tile.hpp; see fit_tile.hip).
#include "wlsqm_tile.hpp"

namespace wlsqm {

int launch_fit_tile_even(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    const bool gather = p.hoods != nullptr;
    if (gather) return WLSQM_OK;
    // Every other even K up to 64 for order 2 (dense input): the same one-wave shape with the shares padded to the next
    // multiple of 8 slots (KC; the extra slots are masked like those of a ragged case), or two waves x two lanes per case
    // for the large 3D neighbourhoods, or two waves with one lane per case where K/2 is even and small.  tools/tune.py
    // at 1M cases against the runtime-K kernels: 2D K = 20 / 28 / 30 / 36 / 44 / 50 / 52 / 60: 0.118 / 0.164 / 0.184 /
    // 0.209 / 0.257 / 0.296 / 0.287 / 0.326 ms against 0.223 / 0.254 / 0.263 / 0.324 / 0.368 / 0.399 / 0.397 / 0.437;
    // 3D K = 28 / 36 / 44 / 48 / 56 / 64: 0.261 / 0.371 / 0.456 / 0.403 / 0.508 / 0.560 against 0.449 / 0.609 / 0.695 /
    // 0.692 / 0.823 / 0.977.  (Index-based input of these sizes stays on the runtime-K one-wave kernel.)
#define PAD_CASE(D, O, KK, KS, LL, UU)                                                                                   \
    if (!gather && dimension == D && order == O && max_nk == KK) {                                                      \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, KS, LL, UU, 2, false, true, true, false, (KK + 7) / 8 * 8>(p, stream);       \
    }
#define HALF_CASE(D, O, KK)                                                                                               \
    if (!gather && dimension == D && order == O && max_nk == KK) {                                                      \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, 2, 1, 2, 2, false, true, true>(p, stream);                                    \
    }
    PAD_CASE(2, 2, 8, 1, 4, 2) PAD_CASE(2, 2, 10, 1, 4, 4) HALF_CASE(2, 2, 12)
    PAD_CASE(2, 2, 14, 1, 4, 4) PAD_CASE(2, 2, 18, 1, 4, 6) HALF_CASE(2, 2, 20)
    PAD_CASE(2, 2, 22, 1, 4, 6) PAD_CASE(2, 2, 26, 1, 4, 8) HALF_CASE(2, 2, 28)
    PAD_CASE(2, 2, 30, 1, 4, 8) PAD_CASE(2, 2, 34, 1, 4, 10) PAD_CASE(2, 2, 36, 1, 4, 10)
    PAD_CASE(2, 2, 38, 1, 4, 10) PAD_CASE(2, 2, 40, 1, 4, 10) PAD_CASE(2, 2, 42, 1, 4, 12)
    PAD_CASE(2, 2, 44, 1, 4, 12) PAD_CASE(2, 2, 46, 1, 4, 12) PAD_CASE(2, 2, 50, 1, 4, 14)
    PAD_CASE(2, 2, 52, 1, 4, 14) PAD_CASE(2, 2, 54, 1, 4, 14) PAD_CASE(2, 2, 56, 1, 4, 14)
    PAD_CASE(2, 2, 58, 1, 4, 16) PAD_CASE(2, 2, 60, 1, 4, 16) PAD_CASE(2, 2, 62, 1, 4, 16)
    HALF_CASE(3, 2, 12) PAD_CASE(3, 2, 14, 1, 4, 2) PAD_CASE(3, 2, 16, 1, 4, 2)
    PAD_CASE(3, 2, 18, 1, 4, 2) HALF_CASE(3, 2, 20) PAD_CASE(3, 2, 22, 1, 4, 2)
    PAD_CASE(3, 2, 24, 1, 4, 2) PAD_CASE(3, 2, 26, 1, 4, 2) HALF_CASE(3, 2, 28)
    PAD_CASE(3, 2, 30, 1, 4, 2) PAD_CASE(3, 2, 34, 1, 4, 2) PAD_CASE(3, 2, 36, 1, 4, 2)
    PAD_CASE(3, 2, 38, 1, 4, 2) PAD_CASE(3, 2, 42, 1, 4, 2) PAD_CASE(3, 2, 44, 1, 4, 2)
    PAD_CASE(3, 2, 46, 1, 4, 2) PAD_CASE(3, 2, 48, 1, 4, 2) PAD_CASE(3, 2, 50, 2, 2, 2)
    PAD_CASE(3, 2, 52, 2, 2, 2) PAD_CASE(3, 2, 54, 2, 2, 2) PAD_CASE(3, 2, 56, 2, 2, 2)
    PAD_CASE(3, 2, 58, 2, 2, 2) PAD_CASE(3, 2, 60, 2, 2, 2) PAD_CASE(3, 2, 62, 2, 2, 2)
    PAD_CASE(3, 2, 64, 2, 2, 2)
    // The other families (dense input, every even K up to 64), shapes from the same A/B (1M cases, ms per launch, against the
    // better runtime-K kernel): 1D, two waves per 64-case tile, moment form — order 2 at K = 6 / 10 / 20: 0.025 / 0.037 /
    // 0.075 against 0.054 / 0.067 / 0.100, order 4 at K = 12 / 24: 0.050 / 0.102 against 0.096 / 0.144; 2D order 1 and 3D
    // order 1, two waves x two lanes per case, padded shares — 2D K = 10 / 20 / 40: 0.062 / 0.110 / 0.197 against 0.082 /
    // 0.129 / 0.284, 3D K = 16 / 24 / 32 / 40: 0.118 / 0.165 / 0.211 / 0.249 against 0.140 / 0.202 / 0.238 / 0.333; 2D order 3,
    // one wave, padded shares — K = 24 / 32 / 48 / 64: 0.230 / 0.262 / 0.332 / 0.401 against 0.291 / 0.327 / 0.432 / 0.591.
#define DENSE_CASE(D, O, KK, ...)                                                                                         \
    if (!gather && dimension == D && order == O && max_nk == KK) {                                                      \
        *handled = true;                                                                                                \
        return launch_tile_impl<D, O, KK, __VA_ARGS__>(p, stream);                                                      \
    }
#define EVEN_K(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) \
    X(40) X(42) X(44) X(46) X(48) X(50) X(52) X(54) X(56) X(58) X(60) X(62) X(64)
#define LINE_CASES(KK) DENSE_CASE(1, 1, KK, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 2, KK, 2, 1, 4, 2, false, false, true) \
    DENSE_CASE(1, 3, KK, 2, 1, 4, 2, false, false, true) DENSE_CASE(1, 4, KK, 2, 1, 4, 2, false, false, true)
#define ORDER1_CASES(KK) PAD_CASE(2, 1, KK, 2, 2, 2) PAD_CASE(3, 1, KK, 2, 2, 2)
#define CUBIC_CASES(KK) PAD_CASE(2, 3, KK, 1, 4, 2)
    if (dimension == 1 && !(order == 2 && (max_nk == 8 || max_nk == 16))) { EVEN_K(LINE_CASES) }
    if (order == 1 && !(dimension == 2 && (max_nk == 16 || max_nk == 32))) { EVEN_K(ORDER1_CASES) }
    if (dimension == 2 && order == 3 && max_nk >= 10 && max_nk != 40) { EVEN_K(CUBIC_CASES) }
#undef LINE_CASES
#undef ORDER1_CASES
#undef CUBIC_CASES
#undef EVEN_K
#undef DENSE_CASE
#undef PAD_CASE
#undef HALF_CASE
    return WLSQM_OK;
}

}  // namespace wlsqm
