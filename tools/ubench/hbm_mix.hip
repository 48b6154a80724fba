// hbm_mix.hip — what a stream with the fit kernels' read : write mix reaches on this part: a workgroup reads 16 pieces and writes one
// (C2: 804 B read, 48 B written per fit = 16.75 : 1), plain and non-temporal stores, at the launch size of the headline kernel (852 MB
// per launch) and at 4 GiB.  The ceiling the `roofline.frac` of bench.py should be read against besides the 8 TB/s pin rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void mix(const d2* __restrict__ x, d2* __restrict__ y, long long n) {
    // a tile = U * 256 contiguous pieces read, 256 pieces written (one store per lane per U loads); workgroups stride over the tiles
    const long long ntiles = n / (256ll * U);
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const d2* src = x + t * 256ll * U + threadIdx.x;
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[u * 256];
        d2 acc = v[0];
#pragma unroll
        for (int u = 1; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
        if (NT) __builtin_nontemporal_store(acc, &y[t * 256 + threadIdx.x]); else y[t * 256 + threadIdx.x] = acc;
    }
}
int main() {
    const long long cap = 4ll << 30;
    d2 *x, *y;
    hipMalloc(&x, cap); hipMalloc(&y, cap / 16 + (1 << 20));
    hipMemset(x, 0, cap);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto launch, const char* name, double moved) {
        float best = 1e9, sum = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; if (rep >= 5) sum += ms;
        }
        printf("%-64s best %.4f ms %.2f TB/s   mean %.4f ms %.2f TB/s\n", name, best, moved / (best * 1e-3) / 1e12, sum / 15, moved / (sum / 15 * 1e-3) / 1e12);
    };
    for (long long rbytes : {804000000ll, 4ll << 30}) {
        const long long n = rbytes / 16;
        const double moved = (double)n * 16 * (1.0 + 1.0 / 16);
        for (int wg : {2048, 8192, 32768}) {
            char nm[96];
            snprintf(nm, 96, "read %lld MB + write 1/16, plain stores, %d WGs", rbytes / 1000000, wg);
            timeit([&] { hipLaunchKernelGGL((mix<16, false>), dim3(wg), dim3(256), 0, 0, x, y, n); }, nm, moved);
            snprintf(nm, 96, "read %lld MB + write 1/16, non-temporal stores, %d WGs", rbytes / 1000000, wg);
            timeit([&] { hipLaunchKernelGGL((mix<16, true>), dim3(wg), dim3(256), 0, 0, x, y, n); }, nm, moved);
        }
    }
    return 0;
}
