// fp64_issue.hip — how fast ONE wave per SIMD issues independent v_fma_f64, against two / four waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/fp64_issue.hip -o /tmp/fp64_issue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int NACC>
__global__ __launch_bounds__(64) void fma_chain(double* out, int iters, double a, double b) {
    extern __shared__ double lds[];
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) out[threadIdx.x] = s + lds[0];
}
template <int NACC> void run(int waves_per_simd, double* d) {
    const int iters = 2000;
    const size_t lds = 160 * 1024 / (4 * waves_per_simd) - 512;     // caps residency at 4 * waves_per_simd workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(fma_chain<NACC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fma_chain<NACC>, dim3(grid), dim3(64), lds, 0, d, iters, 1.0000001, 1e-9);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 8 * NACC;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("acc %2d  waves/SIMD %d: %.3f ms  -> %.2f cycles (at 2.4 GHz) per v_fma_f64 per SIMD, %.1f TFLOP/s\n", NACC, waves_per_simd, ms,
           cyc / (instr_per_wave * waves_per_simd), 2.0 * 64 * instr_per_wave * grid / (ms * 1e-3) / 1e12);
}
int main() {
    double* d; hipMalloc(&d, 4096);
    for (int w : {1, 2, 4}) { run<4>(w, d); run<16>(w, d); run<60>(w, d); }
    return 0;
}
