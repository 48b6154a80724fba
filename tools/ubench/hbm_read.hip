// hbm_read.hip — what a read-mostly stream reaches on this part (the fit kernels read ~94 % of their bytes), next to the
// float4 copy figure of MI355X_MICROARCH.md (6.29 TB/s).  Reads 4 GiB in 16-byte pieces, writes one double per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U>
__global__ __launch_bounds__(256) void rd(const d2* __restrict__ x, long long n, double* out) {
    double acc = 0;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    for (; i < n; i += stride) acc += x[i].x + x[i].y;
    if (acc == 1.2345e300) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void cp(const d2* __restrict__ x, d2* __restrict__ y, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = x[i];
}
int main() {
    const long long bytes = 4ll << 30, n = bytes / 16;
    d2 *x, *y; double* o;
    hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&o, 1 << 20);
    hipMemset(x, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto launch, const char* name, double moved) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-34s %.3f ms  %.2f TB/s\n", name, best, moved / (best * 1e-3) / 1e12);
    };
    for (int wg : {2048, 8192, 32768}) {
        char nm[64];
        snprintf(nm, 64, "read 16 B/lane x4 in flight, %d WGs", wg);
        timeit([&] { hipLaunchKernelGGL(rd<4>, dim3(wg), dim3(256), 0, 0, x, n, o); }, nm, (double)bytes);
        snprintf(nm, 64, "read 16 B/lane x8 in flight, %d WGs", wg);
        timeit([&] { hipLaunchKernelGGL(rd<8>, dim3(wg), dim3(256), 0, 0, x, n, o); }, nm, (double)bytes);
    }
    timeit([&] { hipLaunchKernelGGL(cp, dim3(16384), dim3(256), 0, 0, x, y, n); }, "copy 16 B/lane (read + write)", 2.0 * bytes);
    return 0;
}
