// exact_div_sqrt.hip — the accurate mode's quotient and root sequences (csrc/fit_accurate.hip: the instruction sequences hipcc emits
// for `/` and sqrt() on fp64, LLVM AMDGPU LowerFDIV64 / lowerFSQRTF64, WITHOUT v_div_scale / v_div_fmas' scaling / v_div_fixup and
// without sqrt's range scaling and class test) against the compiler's own `/` and sqrt() on random operands in the range the
// kernel admits them for: every magnitude in [2^-200, 2^200] (quotients therefore in [2^-400, 2^400]).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/exact_div_sqrt tools/ubench/exact_div_sqrt.hip && ./tools/ubench/exact_div_sqrt [log2 pairs per class = 31]
// Operand classes: (0) independent mantissas and exponents; (1) a <= b with a shared exponent neighbourhood (the weights' d2 / max_d2);
// (2) quotients within a few ulp of 1 and of powers of two (rounding boundaries); (3) one refined reciprocal shared by many
// numerators (the weights' per-case reciprocal); (4) - (6) the SEEDED quotient (one Newton step from a seed within 2^-44 of the
// reciprocal: random denominators, adversarial denominators, seeds whole ulps off).  Prints mismatch counts (must all be 0) and exits non-zero otherwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

#pragma clang fp contract(off)

__device__ __forceinline__ double rcp_refined(double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0); r = fma(r, e, r);
    e = fma(-b, r, 1.0); r = fma(r, e, r);
    return r;
}
__device__ __forceinline__ double div_by(double a, double b, double r) {
    const double q = a * r;
    const double e = fma(-b, q, a);
    return fma(e, r, q);
}
// one Newton step from a seed that is already within 2^-44 of 1 / b (the equilibration's product of two running reciprocal scale
// factors), then the same quotient / remainder / correction steps: no v_rcp_f64 (a quarter-rate instruction)
__device__ __forceinline__ double div_seeded(double a, double b, double seed) {
    const double e = fma(-b, seed, 1.0);
    const double r = fma(seed, e, seed);
    return div_by(a, b, r);
}
__device__ __forceinline__ double fast_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    double d = fma(-g, g, x); g = fma(d, h, g);
    d = fma(-g, g, x); g = fma(d, h, g);
    return g;
}
__device__ __forceinline__ uint64_t next(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ __forceinline__ double make(uint64_t mant, int exp2, bool neg) {      // 1.mant * 2^exp2
    const uint64_t bits = ((uint64_t)(exp2 + 1023) << 52) | (mant & 0xfffffffffffffull) | (neg ? 0x8000000000000000ull : 0ull);
    return __longlong_as_double((long long)bits);
}

__global__ void check(int cls, uint64_t seed, int per_thread, unsigned long long* bad_div, unsigned long long* bad_sqrt) {
    uint64_t s = seed ^ (0x9e3779b97f4a7c15ull * (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x + 1));
    next(s); next(s);
    unsigned long long bd = 0, bs = 0;
    double shared_b = make(next(s), (int)(next(s) % 401) - 200, false), shared_r = rcp_refined(shared_b);
    for (int it = 0; it < per_thread; ++it) {
        double a, b;
        const uint64_t m1 = next(s), m2 = next(s), e = next(s);
        if (cls == 0 || cls >= 4) {
            a = make(m1, (int)(e % 401) - 200, (e >> 20) & 1); b = make(m2, (int)((e >> 32) % 401) - 200, (e >> 21) & 1);
            if (cls >= 4) { a = fabs(a); b = fabs(b); }
        } else if (cls == 1) {
            b = make(m2, (int)(e % 401) - 200, false); a = b * ((double)(m1 >> 11) * 0x1p-53);       // 0 <= a <= b
            if (a < 0x1p-200) a = b;
        } else if (cls == 2) {
            b = make(m2, (int)(e % 401) - 200, false);
            const int k = (int)((e >> 32) % 9) - 4, sh = (int)((e >> 40) % 5) - 2;                // a = b * 2^sh * (1 + k ulp)
            a = __longlong_as_double(__double_as_longlong(ldexp(b, sh)) + k);
        } else {
            b = shared_b; a = b * ((double)(m1 >> 11) * 0x1p-53);
            if (a < 0x1p-200) a = b;
        }
        const double q_ref = a / b;
        double q = (cls == 3) ? div_by(a, b, shared_r) : div_by(a, b, rcp_refined(b));
        if (cls >= 4) {
            // seeds: the correctly rounded reciprocal (1.0 / b) pushed off by up to +-2^-44 relative (classes 4, 5) or by whole ulps (6)
            const double r0 = 1.0 / b;
            double seed;
            if (cls == 6) seed = __longlong_as_double(__double_as_longlong(r0) + (long long)((e >> 48) % 513) - 256);
            else seed = r0 * (1.0 + ((double)((long long)(m1 >> 12) - (1ll << 51)) * 0x1p-51) * 0x1p-44);
            if (cls == 5) {                                        // adversarial denominators: mantissa all ones / one / powers of two +- ulps
                const int k = (int)((e >> 40) % 7) - 3;
                const uint64_t mant = ((e >> 44) & 1) ? 0xfffffffffffffull : 0ull;
                b = __longlong_as_double(__double_as_longlong(make(mant, (int)(e % 401) - 200, false)) + k);
                const double r1 = 1.0 / b;
                seed = r1 * (1.0 + ((double)((long long)(m1 >> 12) - (1ll << 51)) * 0x1p-51) * 0x1p-44);
            }
            q = div_seeded(a, b, seed);
            bd += (__double_as_longlong(q) != __double_as_longlong(a / b));
            continue;
        }
        bd += (__double_as_longlong(q) != __double_as_longlong(q_ref));
        const double x = fabs(cls == 0 ? a : q_ref);                                              // roots of operands and of quotients (the weights' argument)
        if (x >= 0x1p-400 && x <= 0x1p400) {
            const double r_ref = sqrt(x), r = fast_sqrt(x);
            bs += (__double_as_longlong(r) != __double_as_longlong(r_ref));
        }
    }
    if (bd) atomicAdd(bad_div, bd);
    if (bs) atomicAdd(bad_sqrt, bs);
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 31;
    unsigned long long *d_bad, h_bad[2];
    if (hipMalloc(&d_bad, 16) != hipSuccess) { printf("no device\n"); return 2; }
    const int threads = 256, blocks = 4096, per_thread = (int)((1ull << lg) / ((unsigned long long)threads * blocks));
    int rc = 0;
    for (int cls = 0; cls < 7; ++cls) {
        hipMemset(d_bad, 0, 16);
        hipLaunchKernelGGL(check, dim3(blocks), dim3(threads), 0, 0, cls, 0x1234567ull + 977 * cls, per_thread, d_bad, d_bad + 1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
        hipMemcpy(h_bad, d_bad, 16, hipMemcpyDeviceToHost);
        printf("class %d: %llu operand pairs: quotient mismatches %llu, root mismatches %llu\n", cls,
               (unsigned long long)threads * blocks * per_thread, h_bad[0], h_bad[1]);
        if (h_bad[0] || h_bad[1]) rc = 1;
    }
    printf(rc ? "FAILED\n" : "all sequences bit-identical to the compiler's IEEE `/` and sqrt()\n");
    return rc;
}
