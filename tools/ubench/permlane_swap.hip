// permlane_swap.hip — what v_permlane16_swap / v_permlane32_swap (gfx950) do to their two operands, measured.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned* out) {
    const unsigned l = threadIdx.x;
    const unsigned x = 1000 + l, y = 2000 + l;
    const u2 a = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    const u2 b = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[l * 4 + 0] = a.x; out[l * 4 + 1] = a.y; out[l * 4 + 2] = b.x; out[l * 4 + 3] = b.y;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 4 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 5, 16, 21, 32, 37, 48, 53})
        printf("lane %2d (x = 1000 + l, y = 2000 + l): permlane16_swap -> (%u, %u)   permlane32_swap -> (%u, %u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
