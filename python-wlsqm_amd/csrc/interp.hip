// interp.hip — evaluation of fitted local models and their derivatives at arbitrary points:
// interp.interpolate_{1,2,3}D (interp.pyx:252-935) over polyeval.taylor_*/general_* (polyeval.pyx),
// as used by interpolate_fit (interp.pyx:34-143) and ExpertSolver.interpolate (expert.pyx:687-985).
//
// The reference hard-codes, per (dimension, diff), the shifted coefficient vector of the derivative
// polynomial and evaluates it in Horner form.  Here one formula covers every case: with P_a the
// exponent multi-index of DOF a (defs.pyx:91-183) and Q that of `diff`,
//     d^Q model (x) = sum_{a : P_a >= Q} fi[a] * prod_m (x_m - xi_m)^(P_a - Q)_m / (P_a - Q)_m!
// and diff >= no gives 0 (interp.pyx:674-678).  One lane per evaluation point; a streaming kernel
// (8*dim bytes in, 8 out per point, plus the gathered model row).
#include <vector>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_interp.hpp"

namespace wlsqm {

template <int DIM>
__global__ __launch_bounds__(256) void interp_kernel(const InterpParams q) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= q.nx) return;
    double xp[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) xp[c] = q.x[m * q.sx + c];
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    if (q.list_off) {                                   // continuous: weighted average of the models within r
        double acc = 0.0, sum_w = 0.0;
        for (long long e = q.list_off[m]; e < q.list_off[m + 1]; ++e) {
            double d2;
            const double v = eval_model<DIM>(q, q.list_idx[e], xp, &d2);
            const double t = 1.0 - sqrt(d2 / q.r2);     // expert.pyx:45-46: alpha = 0, beta = 1
            const double w = t * t;
            acc += w * v; sum_w += w;
        }
        q.out[m] = acc / sum_w;                         // empty list: 0/0 = NaN, as the reference
        return;
    }
    const long long model = q.I ? q.I[m] : (q.nmodels == 1 ? 0 : m);
    q.out[m] = (model < 0 || model >= q.nmodels) ? nan : eval_model<DIM>(q, model, xp, nullptr);
}

int launch_interp(int dimension, const InterpParams& q, hipStream_t stream) {
    if (q.nx <= 0) return WLSQM_OK;
    const long long blocks = (q.nx + 255) / 256;
    if (dimension == 1) hipLaunchKernelGGL(interp_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, q);
    else if (dimension == 2) hipLaunchKernelGGL(interp_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, q);
    else if (dimension == 3) hipLaunchKernelGGL(interp_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, stream, q);
    else { set_error("dimension must be 1, 2 or 3"); return WLSQM_EVALUE; }
    WLSQM_HIP_CHECK(hipGetLastError());
    return WLSQM_OK;
}

}  // namespace wlsqm

using namespace wlsqm;

extern "C" {

// interpolate_fit (interp.pyx:34-143): ONE model (xi[dim], fi[no], order) evaluated at nx host points.
int wlsqm_hip_interpolate_fit_host(int dimension, int order, const double* xi, const double* fi,
                                   const double* x, int64_t x_stride, int64_t nx, int diff, double* out, int device) {
    if (dimension < 1 || dimension > 3) { set_error("dimension must be 1, 2 or 3"); return WLSQM_EVALUE; }
    const int no = wlsqm_hip_number_of_dofs(dimension, order);
    if (no < 0) { set_error("order must be 0, 1, 2, 3 or 4"); return WLSQM_EVALUE; }
    if (!xi || !fi || !x || !out || nx < 0) { set_error("null argument"); return WLSQM_EVALUE; }
    DeviceScope scope; int rc = scope.enter(device);
    if (rc != WLSQM_OK) return rc;
    if (nx == 0) return WLSQM_OK;
    std::vector<double> sx((size_t)nx * dimension);
    for (int64_t m = 0; m < nx; ++m)
        for (int c = 0; c < dimension; ++c) sx[(size_t)m * dimension + c] = x[m * x_stride + c];
    DevBuf d_x, d_xi, d_fi, d_o, d_out;
    if ((rc = d_x.alloc(sx.size() * 8)) || (rc = d_xi.alloc(dimension * 8)) || (rc = d_fi.alloc(no * 8)) ||
        (rc = d_o.alloc(4)) || (rc = d_out.alloc((size_t)nx * 8))) return rc;
    hipStream_t s = nullptr;
    WLSQM_HIP_CHECK(hipMemcpyAsync(d_x.p, sx.data(), d_x.n, hipMemcpyHostToDevice, s));
    WLSQM_HIP_CHECK(hipMemcpyAsync(d_xi.p, xi, d_xi.n, hipMemcpyHostToDevice, s));
    WLSQM_HIP_CHECK(hipMemcpyAsync(d_fi.p, fi, d_fi.n, hipMemcpyHostToDevice, s));
    WLSQM_HIP_CHECK(hipMemcpyAsync(d_o.p, &order, 4, hipMemcpyHostToDevice, s));
    InterpParams q{};
    q.xi = d_xi.as<double>(); q.sxi = dimension; q.fi = d_fi.as<double>(); q.sfi = no;
    q.order = d_o.as<int>(); q.sorder = 0; q.nmodels = 1; q.I = nullptr;
    q.x = d_x.as<double>(); q.sx = dimension; q.nx = nx; q.diff = diff; q.out = d_out.as<double>();
    rc = launch_interp(dimension, q, s);
    if (rc != WLSQM_OK) return rc;
    WLSQM_HIP_CHECK(hipMemcpyAsync(out, d_out.p, (size_t)nx * 8, hipMemcpyDeviceToHost, s));
    WLSQM_HIP_CHECK(hipStreamSynchronize(s));
    return WLSQM_OK;
}

}  // extern "C"
