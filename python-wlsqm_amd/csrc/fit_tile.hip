// fit_tile.hip — dispatch of the fixed-K tile kernels (wlsqm_tile.hpp): eligibility, the two-kernel moment path of 2D
// order 4, the BASELINE shapes with their A/B variants and the curated list; the per-family tables for every even K live in
// fit_tile_even.hip (dense, K <= 64), fit_tile_gather.hip (index-based) and fit_tile_big.hip (64 < K <= 128) so that the
// ~700 instantiations compile in parallel.
#include "wlsqm_tile.hpp"

namespace wlsqm {

// per-family tables (other translation units); *handled stays false when the table has no entry
int launch_fit_tile_even(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled);
int launch_fit_tile_gather(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled);
int launch_fit_tile_big(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled);

// The tile path needs: no extras, all cases in order, dense contiguous arrays, 16-byte aligned bases
// (index-based mode: contiguous 16-byte aligned hoods rows, 16-byte aligned S).
static bool tile_eligible(int dim, const KParams& p, long long K) {
    if (p.do_sens || p.iterative || p.case_index) return false;
    if (p.hoods) {
        if (p.shoods_j != K || (K % 2) != 0) return false;
        return ((reinterpret_cast<uintptr_t>(p.hoods) | reinterpret_cast<uintptr_t>(p.S)) & 15u) == 0;
    }
    if (p.sxk_k != dim || p.sxk_j != K * dim || p.sfk_k != 1 || p.sfk_j != K) return false;
    if ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(p.fk)) & 15u) return false;
    return true;
}

// exported for fit_ring.hip: dense contiguous input the tile kernels can take
bool tile_dense_eligible(int dimension, const KParams& p, long long max_nk) { return !p.hoods && tile_eligible(dimension, p, max_nk); }

// First kernel of the two-kernel moment path (fit_moment.hip): tile pass that leaves the moments in p.ws.
// `handled` stays false when no instantiation covers (dimension, order, max_nk) or the input is not tile-eligible.
// Neighbour-slot counts with a two-kernel moment instantiation (2D order 4).  The host entry points round their device
// rows up to the next of these (preferred_slots), so every host-array batch of 2D order-4 fits with <= 100 neighbours
// takes this path (the reference's own example, examples/wlsqm_example.py:55-187, is order 4 with max_nk = 100).
static bool moment_slots(long long K) { return K >= 16 && K <= 100 && (K % 2) == 0; }      // every even size (shares padded to a multiple of 4)

bool tile_moments_supported(int dimension, int order, const KParams& p, long long max_nk) {
    return dimension == 2 && order == 4 && moment_slots(max_nk) && tile_eligible(dimension, p, max_nk);
}

// Device row length (neighbour slots) the host entry points should allocate for a batch whose largest neighbourhood has
// max_nk members: even (16-byte rows for the tiled kernels), and for 2D order 4 the next size with a moment kernel.
long long preferred_slots(int dimension, int order, long long max_nk) {
    long long K = max_nk + (max_nk & 1);
    if (dimension == 2 && order == 4 && K < 16) K = 16;
    return K < 2 ? 2 : K;
}

int launch_tile_moments(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    if (!tile_moments_supported(dimension, order, p, max_nk)) return WLSQM_OK;
    const bool gather = p.hoods != nullptr;
    *handled = true;
    // A/B at 1M cases, K = 64: two waves per 32-case tile 0.73 ms (with the solve kernel); direct fk the same; one wave per
    // 16-case tile spills (60 accumulators + 16 fk values per lane) 0.85 ms; (2 or 4 waves) x 4 lanes per case 0.85 ms.
    // The tile kernel holds 256 VGPRs (two waves per SIMD by registers, 1.5 by its 50 KB of LDS; PMC at K = 64: VALU busy
    // 44 %, waves waiting 34 %): keeping the NEXT tile's loads in registers across the tile (issued after the staging
    // barrier) spills 408 B per lane and takes 1.34 instead of 0.70 ms; in a lone-wave build (no spills) it still loses: 0.786
    // against 0.652 ms at K = 64, 1.466 against 1.168 at K = 100, 0.611 against 0.621 at K = 40.
    // Shape sweep under the oversubscribed grids (1M cases, tile + solve kernel): K = 64: two waves x two lanes per case 0.640 ms
    // (unroll 2 / 8: 0.660 / 0.640), two waves x four lanes 0.790, four waves x two 0.788, four waves x one 0.762, direct fk 0.775;
    // K = 32: two waves x one lane per case on 64-case tiles 0.427 against 0.463 (one wave x two or four lanes: 0.615 / 0.665),
    // so the multiples of 4 up to 32 take that shape (K = 16 / 20 / 24 / 28 / 32: 0.335 / 0.512 / 0.364 / 0.582 / 0.417 against 0.380 /
    // 0.663 / 0.532 / 0.616 / 0.459; at K = 22 / 30 it loses, 0.509 / 0.615 against 0.426 / 0.462).
    // All of them are compiled for a LONE wave per SIMD (__launch_bounds__(128, 1)): their 50-80 KB of LDS allow 1.5 waves per
    // SIMD at best, and planned for two the 60 accumulators spill 50-580 B per lane at most sizes (K = 64 happens not to) — 400k
    // cases at K = 20 / 28 / 36 / 40 / 56 / 80 / 100: 0.178 / 0.198 / 0.283 / 0.274 / 0.317 / 0.362 / 0.502 ms against 0.228 / 0.266 /
    // 0.381 / 0.324 / 0.396 / 0.404 / 0.991; K = 24 / 48 / 64, which did not spill: unchanged.  Index-based input at K = 40 / 64 / 100:
    // 0.275 / 0.270 / 0.453 against 0.278 / 0.267 / 0.699.
#define MOMENT_CASE(KK)                                                                                             \
    if (max_nk == KK) {                                                                                             \
        if (gather) return launch_tile_impl<2, 4, KK, 2, 2, 4, 1, true, false, true, true, (KK + 3) / 4 * 4>(p, stream);   \
        if constexpr (KK <= 32 && KK % 4 == 0)                                                                      \
            return launch_tile_impl<2, 4, KK, 2, 1, 4, 1, false, false, true, true>(p, stream);                     \
        else                                                                                                        \
            return launch_tile_impl<2, 4, KK, 2, 2, 4, 1, false, false, true, true, (KK + 3) / 4 * 4>(p, stream);  \
    }
    MOMENT_CASE(16) MOMENT_CASE(18) MOMENT_CASE(20) MOMENT_CASE(22) MOMENT_CASE(24) MOMENT_CASE(26) MOMENT_CASE(28) MOMENT_CASE(30)
    MOMENT_CASE(32) MOMENT_CASE(34) MOMENT_CASE(36) MOMENT_CASE(38) MOMENT_CASE(40) MOMENT_CASE(42) MOMENT_CASE(44) MOMENT_CASE(46)
    MOMENT_CASE(48) MOMENT_CASE(50) MOMENT_CASE(52) MOMENT_CASE(54) MOMENT_CASE(56) MOMENT_CASE(58) MOMENT_CASE(60) MOMENT_CASE(62)
    MOMENT_CASE(64) MOMENT_CASE(66) MOMENT_CASE(68) MOMENT_CASE(70) MOMENT_CASE(72) MOMENT_CASE(74) MOMENT_CASE(76) MOMENT_CASE(78)
    MOMENT_CASE(80) MOMENT_CASE(82) MOMENT_CASE(84) MOMENT_CASE(86) MOMENT_CASE(88) MOMENT_CASE(90) MOMENT_CASE(92) MOMENT_CASE(94)
    MOMENT_CASE(96) MOMENT_CASE(98) MOMENT_CASE(100)
#undef MOMENT_CASE
    *handled = false;
    return WLSQM_OK;
}

int launch_fit_tile(int dimension, int order, const KParams& p, long long max_nk, hipStream_t stream, bool* handled) {
    *handled = false;
    // WLSQM_HIP_DISABLE_TILE=1 forces the generic kernels (A/B measurements and the tile-vs-lane parity test)
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    const char* nofix = getenv("WLSQM_HIP_DISABLE_FIXEDK");     // A/B: send the curated shapes to the runtime-K kernels too
    if (nofix && nofix[0] == '1' && !p.hoods) return WLSQM_OK;
    if (!tile_eligible(dimension, p, max_nk)) return WLSQM_OK;
    // WLSQM_TILE_VARIANT selects a tuning variant of the BASELINE configs (tools/tune.py); default = best measured
    const char* v = getenv("WLSQM_TILE_VARIANT");
    const int var = v ? atoi(v) : 0;
    const bool gather = p.hoods != nullptr;
#define TILE_CASE(D, O, KK, ...)                                                 \
    if (dimension == D && order == O && max_nk == KK) { *handled = true; return launch_tile_any<D, O, KK, __VA_ARGS__>(p, stream, gather); }
    if (dimension == 2 && order == 2 && max_nk == 32) {      // C2
        *handled = true;
        if (gather) {
            // index-based input has no direct-fk option (F is gathered too), and without it the one-wave tiles lose
            // (0.205 vs 0.170 ms per 1M cases): keep four waves per 64-case tile
            return var == 3 ? launch_tile_impl<2, 2, 32, 4, 1, 8, 2, true>(p, stream)
                            : launch_tile_impl<2, 2, 32, 4, 1, 8, 2, true, false, true>(p, stream);
        }
        // Since the grids launch 16 workgroups per resident slot (wlsqm_internal.hpp) one wave per 32-case tile with two
        // lanes per case leads: 0.159 against 0.166 ms for four lanes per case (variant 4) at 1M cases, 0.314 against 0.348 at
        // 2M (228 VGPRs, two waves per SIMD; unroll 4 / 16: 0.159 / 0.158 against 0.157; squeezed to 168 VGPRs for three waves,
        // by unroll 2 or __launch_bounds__(64, 3): 0.169 / 0.190).  Earlier A/B at 1M cases with resident-size grids (tools/tune.py), ms per launch: one wave per 16-case tile + moments + direct fk 0.167;
        // the same with two lanes per case 0.173; without direct fk 0.233; four waves per 64-case tile: moments
        // 0.181, entry form 0.186 (the round-1 kernel); eight waves 0.43; the default shape squeezed to 128 VGPRs (four
        // waves per SIMD, small spills) 0.182-0.199, with unroll 4 or 2 at three waves per SIMD 0.176-0.180.
        switch (var) {
            case 2: return launch_tile_impl<2, 2, 32, 4, 1, 4, 3, false, false, true>(p, stream);
            case 3: return launch_tile_any<2, 2, 32, 4, 1, 8, 2>(p, stream, gather);
            case 4: return launch_tile_impl<2, 2, 32, 1, 4, 8, 2, false, true, true>(p, stream);
            default: return launch_tile_impl<2, 2, 32, 1, 2, 8, 2, false, true, true>(p, stream);
        }
    }
    if (dimension == 3 && order == 2 && max_nk == 40) {      // C5
        *handled = true;
        // A/B at 1M cases, ms per launch: one wave per 16-case tile + moments + direct fk 0.349; two waves per
        // 32-case tile: moments + direct fk 0.359, moments 0.395, entry form + direct fk 0.477, entry form 0.567 (the
        // round-1 kernel); four waves (4 x 1 lanes per case) spill.
        switch (var) {
            case 1: return launch_tile_impl<3, 2, 40, 2, 2, 2, 2, false, true, true>(p, stream);
            case 2: return launch_tile_impl<3, 2, 40, 1, 4, 10, 2, false, true, true>(p, stream);
            case 3: return launch_tile_any<3, 2, 40, 2, 2, 2, 2>(p, stream, gather);
            default: return launch_tile_any<3, 2, 40, 1, 4, 2, 2, true, true>(p, stream, gather);
        }
    }
    // The per-family tables first (every even K up to 128; fit_tile_even.hip / fit_tile_gather.hip / fit_tile_big.hip), then
    // the shapes curated one by one: best of {entry form, moment form} x {four waves per 64-case tile, two waves per 32, one wave
    // per 16 cases with direct fk}, tools/tune.py at 1M cases; anything else -> fit_tilek.hip.
    {
        int rc = max_nk > 64 ? launch_fit_tile_big(dimension, order, p, max_nk, stream, handled)
                 : gather    ? launch_fit_tile_gather(dimension, order, p, max_nk, stream, handled)
                             : launch_fit_tile_even(dimension, order, p, max_nk, stream, handled);
        if (rc != WLSQM_OK || *handled) return rc;
    }
    TILE_CASE(2, 2, 16, 1, 4, 4, 2, true, true)
    TILE_CASE(2, 2, 24, 2, 1, 4, 2, false, true)
    TILE_CASE(2, 2, 48, 4, 1, 4, 2, false, true)
    TILE_CASE(2, 2, 64, 1, 4, 4, 2, true, true)
    TILE_CASE(2, 1, 16, 4, 1, 4, 2)
    TILE_CASE(2, 1, 32, 4, 1, 4, 2)
    TILE_CASE(1, 2, 8, 2, 1, 4, 2)
    TILE_CASE(1, 2, 16, 1, 4, 4, 2, true, true)
    TILE_CASE(3, 1, 32, 4, 1, 4, 2)
    TILE_CASE(3, 2, 32, 1, 4, 2, 2, true, true)
    TILE_CASE(2, 3, 40, 2, 2, 2, 2, true, true)
#undef TILE_CASE
    return WLSQM_OK;
}

}  // namespace wlsqm
