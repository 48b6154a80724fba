// mfma_f64_layout.hip — register layout and issue rate of v_mfma_f64_16x16x4_f64 on gfx950, measured.
// build: hipcc -w --offload-arch=gfx950 -O3 tools/ubench/mfma_f64_layout.hip -o tools/ubench/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    const int i = l % 16, k = l / 16;
    const double a = (i + 1) + 0.01 * k;           // assumed: A[i = l % 16][k = l / 16]
    const double b = (i + 1) * 1000.0 + k;         // assumed: B[k = l / 16][j = l % 16]
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
__global__ __launch_bounds__(64) void rate(double* out, int iters) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    d4 s = c0 + c1 + c2 + c3;
    if (s[0] == 12345.6789) out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
    double* d; hipMalloc(&d, 64 * 4 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // reference P[i][j] = sum_k A[i][k] B[k][j]
    int where_i[64][4], where_j[64][4], found = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        where_i[l][v] = where_j[l][v] = -1;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double p = 0; for (int k = 0; k < 4; ++k) p += ((i + 1) + 0.01 * k) * ((j + 1) * 1000.0 + k);
            if (fabs(p - h[l * 4 + v]) < 1e-9 * fabs(p)) { where_i[l][v] = i; where_j[l][v] = j; ++found; }
        }
    }
    printf("entries matched: %d of 256\n", found);
    for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) {
        printf("lane %2d:", l);
        for (int v = 0; v < 4; ++v) printf("  v%d -> D[%d][%d]", v, where_i[l][v], where_j[l][v]);
        printf("\n");
    }
    // issue rate: 4 independent accumulators, one wave per SIMD, then 2 and 4 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w : {1, 2, 4}) {
        const int iters = 20000, grid = 256 * 4 * w;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(rate, dim3(grid), dim3(64), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double n = (double)iters * 4;
        printf("waves/SIMD %d: %.3f ms -> %.1f cycles (at 2.4 GHz) per MFMA per SIMD, %.1f TFLOP/s\n", w, ms,
               ms * 1e-3 * 2.4e9 / (n * w), 2048.0 * n * grid / (ms * 1e-3) / 1e12);
    }
    return 0;
}
