// dpp_newbcast.hip — gfx950: a 64-bit VALU operation with a DPP row_newbcast source (v_fmac_f64_dpp, v_mov_b64_dpp; v_mul_f64 / v_fma_f64 are VOP3: no DPP form).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_newbcast tools/ubench/dpp_newbcast.hip && /tmp/dpp_newbcast
// Checks (1) what lane the broadcast operand comes from (lane N of the reader's own row of 16 lanes), (2) that it is the DPP operand
// (src0) that is permuted, (3) whether a value written by the instruction right before can be read through DPP without wait states
// in between (hand-written inline assembly is invisible to the compiler's hazard recogniser).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int N>
__global__ void k(double* out, const double* in) {
    const int l = threadIdx.x;
    double a = in[l], b = in[64 + l], c = in[128 + l], m, mv, h = in[192 + l];
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(b), "n"(N));      // c += a[row lane N] * b
    m = 0.0;                                                                                                                       // (v_mul_f64 has no DPP form — VOP3 only; a product is an fmac into zero)
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(a), "v"(b), "n"(N));       // m = a[row lane N] * b
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(mv) : "v"(a), "n"(N));
    // hazard probe: h2 is written by a plain VALU instruction and read through DPP by the very next one
    double h2, hz;
    asm volatile("v_add_f64 %0, %2, %2\n\tv_mov_b64_dpp %1, %0 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=&v"(h2), "=v"(hz) : "v"(h), "n"(N));
    out[l] = c; out[64 + l] = m; out[128 + l] = mv; out[192 + l] = hz;
}
int main() {
    double h_in[256], h_out[256], *d_in, *d_out;
    for (int i = 0; i < 256; ++i) h_in[i] = 1.0 + 0.001 * i + (i % 7) * 0.37;
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, sizeof h_out);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    int bad = 0;
    auto check = [&](int N) {
        for (int l = 0; l < 64; ++l) {
            const int src = (l / 16) * 16 + N;
            const double a = h_in[src], b = h_in[64 + l];
            if (h_out[l] != std::fma(a, b, h_in[128 + l])) { ++bad; if (bad < 5) printf("fmac lane %d N %d: %.17g vs %.17g\n", l, N, h_out[l], std::fma(a, b, h_in[128 + l])); }
            if (h_out[64 + l] != a * b) { ++bad; if (bad < 5) printf("mul lane %d N %d\n", l, N); }
            if (h_out[128 + l] != a) { ++bad; if (bad < 5) printf("mov lane %d N %d\n", l, N); }
            if (h_out[192 + l] != 2.0 * h_in[192 + src]) { ++bad; if (bad < 5) printf("hazard lane %d N %d: %.17g vs %.17g\n", l, N, h_out[192 + l], 2.0 * h_in[192 + src]); }
        }
    };
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d_out, d_in); hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost); check(3);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d_out, d_in); hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost); check(0);
    hipLaunchKernelGGL(k<15>, dim3(1), dim3(64), 0, 0, d_out, d_in); hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost); check(15);
    printf("row_newbcast on 64-bit fmac / mul / mov: %s (%d mismatches; the last probe reads a register written by the instruction before it)\n", bad ? "MISMATCH" : "as expected", bad);
    return bad != 0;
}
