// wlsqm_interp.hpp — parameter block of the model-evaluation kernel (interp.hip), shared with expert.hip.
#pragma once
#include <hip/hip_runtime.h>

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

}  // namespace wlsqm
