// fit_tile.hip — contiguous fast path (LDS-staged tiles).  Placeholder dispatcher until the
// tile kernels land: reports "not handled" so launch_fit() falls through to fit_lane / fit_wave.
#include "wlsqm_internal.hpp"

namespace wlsqm {

int launch_fit_tile(int, int, const KParams&, long long, hipStream_t, bool* handled) {
    *handled = false;
    return WLSQM_OK;
}

}  // namespace wlsqm
