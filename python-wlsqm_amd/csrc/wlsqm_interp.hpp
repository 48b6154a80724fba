// wlsqm_interp.hpp — parameter block of the model-evaluation kernel (interp.hip), shared with expert.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "wlsqm_kernels.hpp"

namespace wlsqm {

struct InterpParams {
    const double* xi;  long long sxi;       // model origins [nmodels, dim]
    const double* fi;  long long sfi;       // model coefficients [nmodels, >= no]
    const int* order;  long long sorder;    // per-model order (sorder == 0: one value for all)
    long long nmodels;
    const long long* I;                     // model used for point m (nullable: model 0 when nmodels == 1, else m)
    const double* x;   long long sx;        // evaluation points [nx, dim]
    long long nx;
    int diff;
    double* out;
    // continuous mode (expert.pyx:898-985): CSR lists of the models within radius r of each point
    const long long* list_off; const long long* list_idx; double r2;
};

int launch_interp(int dimension, const InterpParams& q, hipStream_t stream);

// Continuous mode without lists (knn.hip): every evaluation point walks the cells of a uniform grid over the model origins
// that its ball of radius r touches and averages the models inside.  q.I / q.list_* are ignored.
int interp_continuous(int dimension, const InterpParams& q, double r, hipStream_t stream);

// value (or derivative q.diff) of model `model` at xp; *d2_out (nullable) receives |xp - xi[model]|^2
template <int DIM>
__device__ __forceinline__ double eval_model(const InterpParams& q, long long model, const double (&xp)[DIM], double* d2_out) {
    const int order = q.order[model * q.sorder];
    const int no = ndofs(DIM, order);
    double pw[DIM][5];
    double d2 = 0.0;
#pragma unroll
    for (int m = 0; m < DIM; ++m) {
        const double d = xp[m] - q.xi[model * q.sxi + m];
        d2 += d * d;
        pw[m][0] = 1.0; pw[m][1] = d; pw[m][2] = 0.5 * d * d; pw[m][3] = (1.0 / 6.0) * d * d * d;
        pw[m][4] = (1.0 / 24.0) * (d * d) * (d * d);
    }
    if (d2_out) *d2_out = d2;
    if (q.diff >= no || q.diff < 0) return 0.0;
    const int Qx = Mono<DIM>::P[q.diff], Qy = Mono<DIM>::Q[q.diff], Qz = Mono<DIM>::R[q.diff];
    const double* f = q.fi + model * q.sfi;
    double acc = 0.0;
    for (int a = 0; a < no; ++a) {
        const int ex = Mono<DIM>::P[a] - Qx, ey = Mono<DIM>::Q[a] - Qy, ez = Mono<DIM>::R[a] - Qz;
        if (ex < 0 || ey < 0 || ez < 0) continue;
        double term = f[a] * pw[0][ex];
        if constexpr (DIM >= 2) term *= pw[1][ey];
        if constexpr (DIM == 3) term *= pw[2][ez];
        acc += term;
    }
    return acc;
}


}  // namespace wlsqm
