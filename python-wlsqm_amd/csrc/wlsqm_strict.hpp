#pragma once
// wlsqm_strict.hpp — the reference-order building blocks shared by the strict kernels (fit_strict.hip) and the accurate mode
// (fit_accurate.hip): constants, row access, make_c_nD with the reference's grouping of every product, the IEEE weight.
// Every translation unit that includes this file must compile with `#pragma clang fp contract(off)` around its kernels: the
// reference is gcc -O2 on x86-64 (no contraction), and bit-identity between the kernels depends on it.
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

#pragma clang fp contract(off)

namespace wlsqm {

namespace strict {

constexpr double onesixth = 1. / 6.;      // impl.pyx:30-31
constexpr double one24th = 1. / 24.;
constexpr double weights_alpha = 1e-4;    // infra.pyx:45-46
constexpr double weights_beta = 1. - 1e-4;
constexpr double ruiz_epsilon = 1e-15;    // lapackdrivers.pyx:87

// Cases per 64-thread workgroup for a system of NO DOFs: the LDS image is slots(NO) doubles per case.
__host__ __device__ constexpr int slots(int NO) { return NO * NO + 10 * NO; }
__host__ __device__ constexpr int lanes_for(int NO) {
    return slots(NO) * 8 * 64 <= 80 * 1024 ? 64 : slots(NO) * 8 * 32 <= 80 * 1024 ? 32 : slots(NO) * 8 * 16 <= 80 * 1024 ? 16 : 8;
}

template <int DIM>
struct Rows {       // row access of one case: dense rows with strides, or index-based (hoods row into the S / F point tables)
    const double* xr; long long sxk_k;
    const double* fr; long long sfk_k;
    const int* hr; const double* S; const double* F;
    __device__ __forceinline__ void offset(int k, const double (&xi)[DIM], double (&d)[DIM]) const {
        const double* q = hr ? S + (long long)hr[k] * DIM : xr + k * sxk_k;
#pragma unroll
        for (int m = 0; m < DIM; ++m) d[m] = q[m] - xi[m];
    }
    __device__ __forceinline__ double value(int k) const { return hr ? F[hr[k]] : fr[k * sfk_k]; }
};

// c[k, :] with the reference's own grouping of every product; returns the squared distance (its w[k] before the weighting).
template <int DIM, int ORDER>
__device__ __forceinline__ double make_c(const double (&d)[DIM], double (&c)[ndofs(DIM, ORDER)]) {
    if constexpr (DIM == 1) {                                   // impl.pyx:449-544
        const double dx = d[0], dx2 = dx * dx;
        c[0] = 1.;
        if constexpr (ORDER >= 1) c[1] = dx;
        if constexpr (ORDER >= 2) c[2] = 0.5 * dx2;
        if constexpr (ORDER >= 3) c[3] = onesixth * dx * dx2;
        if constexpr (ORDER >= 4) c[4] = one24th * dx2 * dx2;
        return dx2;
    } else if constexpr (DIM == 2) {                            // impl.pyx:286-432
        const double dx = d[0], dy = d[1];
        const double dx2 = dx * dx, dy2 = dy * dy;
        const double d2 = dx2 + dy2;
        c[0] = 1.;
        if constexpr (ORDER >= 1) { c[1] = dx; c[2] = dy; }
        if constexpr (ORDER >= 2) { c[3] = 0.5 * dx2; c[4] = dx * dy; c[5] = 0.5 * dy2; }
        if constexpr (ORDER == 3) {
            c[6] = onesixth * dx2 * dx; c[7] = 0.5 * dx2 * dy; c[8] = 0.5 * dx * dy2; c[9] = onesixth * dy * dy2;
        }
        if constexpr (ORDER == 4) {
            const double dx3 = dx2 * dx, dy3 = dy2 * dy;
            c[6] = onesixth * dx3; c[7] = 0.5 * dx2 * dy; c[8] = 0.5 * dx * dy2; c[9] = onesixth * dy3;
            c[10] = one24th * dx2 * dx2; c[11] = onesixth * dx3 * dy; c[12] = 0.25 * dx2 * dy2; c[13] = onesixth * dx * dy3;
            c[14] = one24th * dy2 * dy2;
        }
        return d2;
    } else {                                                    // impl.pyx:70-269, DOF order defs.pyx:137-171
        const double dx = d[0], dy = d[1], dz = d[2];
        const double dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
        const double d2 = dx2 + dy2 + dz2;
        c[0] = 1.;
        if constexpr (ORDER >= 1) { c[1] = dx; c[2] = dy; c[3] = dz; }
        if constexpr (ORDER >= 2) {
            c[4] = 0.5 * dx2; c[5] = dx * dy; c[6] = 0.5 * dy2; c[7] = dy * dz; c[8] = 0.5 * dz2; c[9] = dx * dz;
        }
        if constexpr (ORDER == 3) {
            c[10] = onesixth * dx2 * dx; c[11] = 0.5 * dx2 * dy; c[12] = 0.5 * dx * dy2; c[13] = onesixth * dy * dy2;
            c[14] = 0.5 * dy2 * dz; c[15] = 0.5 * dy * dz2; c[16] = onesixth * dz * dz2; c[17] = 0.5 * dx * dz2;
            c[18] = 0.5 * dx2 * dz; c[19] = dx * dy * dz;
        }
        if constexpr (ORDER == 4) {
            const double dx3 = dx2 * dx, dy3 = dy2 * dy, dz3 = dz2 * dz;
            c[10] = onesixth * dx3; c[11] = 0.5 * dx2 * dy; c[12] = 0.5 * dx * dy2; c[13] = onesixth * dy3;
            c[14] = 0.5 * dy2 * dz; c[15] = 0.5 * dy * dz2; c[16] = onesixth * dz3; c[17] = 0.5 * dx * dz2;
            c[18] = 0.5 * dx2 * dz; c[19] = dx * dy * dz;
            c[20] = one24th * dx2 * dx2; c[21] = onesixth * dx3 * dy; c[22] = 0.25 * dx2 * dy2; c[23] = onesixth * dx * dy3;
            c[24] = one24th * dy2 * dy2; c[25] = onesixth * dy3 * dz; c[26] = 0.25 * dy2 * dz2; c[27] = onesixth * dy * dz3;
            c[28] = one24th * dz2 * dz2; c[29] = onesixth * dx * dz3; c[30] = 0.25 * dx2 * dz2; c[31] = onesixth * dx3 * dz;
            c[32] = 0.5 * dx2 * dy * dz; c[33] = 0.5 * dx * dy2 * dz; c[34] = 0.5 * dx * dy * dz2;
        }
        return d2;
    }
}

// infra.pyx:668-702: IEEE quotient and IEEE root (hipcc expands both to correctly rounded sequences)
__device__ __forceinline__ double make_weight(double d2, double max_d2, bool uniform) {
    if (uniform) return 1.;
    const double tmp = 1. - sqrt(d2 / max_d2);
    return weights_alpha + weights_beta * tmp * tmp;
}

// c[a] for a per-lane index, registers only: a binary tree of selects on the bits of a.  (A chain of `a == q ? c[q] : r` is
// turned into an indexed load from a scratch copy of c by the optimizer.)
template <int N>
__device__ __forceinline__ double pick(const double (&c)[N], int a) {
    constexpr int P = N <= 1 ? 1 : N <= 2 ? 2 : N <= 4 ? 4 : N <= 8 ? 8 : N <= 16 ? 16 : N <= 32 ? 32 : 64;
    long long t[P];
#pragma unroll
    for (int q = 0; q < P; ++q) t[q] = __double_as_longlong(c[q < N ? q : N - 1]);
#pragma unroll
    for (int w = P / 2, bit = 0; w >= 1; w >>= 1, ++bit) {
        const bool odd = (a >> bit) & 1;
#pragma unroll
        for (int q = 0; q < w; ++q) t[q] = odd ? t[2 * q + 1] : t[2 * q];
    }
    return __longlong_as_double(t[0]);
}


}  // namespace strict

// ---- correctly rounded quotients and roots without the range machinery (round 4; fit_accurate.hip's header has the story) ----
namespace acc {

// ---- correctly rounded quotient and root for in-range operands: the instruction sequences hipcc emits for `/` and sqrt()
// (LLVM AMDGPU LowerFDIV64 / lowerFSQRTF64) without the range scaling and the special-case fix-up
__device__ __forceinline__ double rcp_refined(double b) {          // v_rcp_f64 + two Newton steps
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0); r = fma(r, e, r);
    e = fma(-b, r, 1.0); r = fma(r, e, r);
    return r;
}
__device__ __forceinline__ double div_by(double a, double b, double r) {   // r = rcp_refined(b)
    const double q = a * r;
    const double e = fma(-b, q, a);
    return fma(e, r, q);
}
struct FastOps {
    // (no operand of these is ever a NaN on the fast path — the range checks — so v_max_f64 is the reference's `if (q > acc) acc = q`)
    static __device__ __forceinline__ double maxnum(double acc, double q) { return __builtin_fmax(acc, q); }
    static __device__ __forceinline__ double div(double a, double b) { return div_by(a, b, rcp_refined(b)); }
    // the same quotient from a SEED that is already within 2^-44 of 1 / b: one Newton step instead of v_rcp_f64 (a quarter-rate
    // instruction: 16 cycles) and two.  The refined reciprocal is as accurate as the standard sequence's (its error is the square of
    // the seed's plus one rounding), and the quotient / remainder / correction steps are the same: bit-identical to `/` on 3 x 2^31
    // operand pairs with seeds up to 256 ulps off and adversarial denominators (tools/ubench/exact_div_sqrt.hip, classes 4-6).
    static __device__ __forceinline__ double div_seeded(double a, double b, double seed) {
        const double e = fma(-b, seed, 1.0);
        return div_by(a, b, fma(seed, e, seed));
    }
    static __device__ __forceinline__ double rcp_of(double b) { return rcp_refined(b); }
    static __device__ __forceinline__ double div_r(double a, double b, double r) { return div_by(a, b, r); }
    static __device__ __forceinline__ double sqrt(double x) { double h; return sqrt_h(x, h); }
    // the root, and the refined half reciprocal root h = 1 / (2 sqrt(x)) (1 + O(2^-46)) the sequence computes on the way:
    // 2 h is a seed for the quotient by the root (div_seeded)
    static __device__ __forceinline__ double sqrt_h(double x, double& h) {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y; h = y * 0.5;
        const double r = fma(-h, g, 0.5);
        g = fma(g, r, g); h = fma(h, r, h);
        double d = fma(-g, g, x); g = fma(d, h, g);
        d = fma(-g, g, x); g = fma(d, h, g);
        return g;
    }
    static __device__ __forceinline__ double div_by_root(double a, double s, double h) { return div_seeded(a, s, h + h); }
};
struct IeeeOps {                                                     // the compiler's sequences: any operand
    static __device__ __forceinline__ double maxnum(double acc, double q) { return q > acc ? q : acc; }      // NaN q: skipped; NaN acc: kept
    static __device__ __forceinline__ double div(double a, double b) { return a / b; }
    static __device__ __forceinline__ double div_seeded(double a, double b, double) { return a / b; }
    static __device__ __forceinline__ double rcp_of(double) { return 0.; }
    static __device__ __forceinline__ double div_r(double a, double b, double) { return a / b; }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double sqrt_h(double x, double& h) { h = 0.; return ::sqrt(x); }
    static __device__ __forceinline__ double div_by_root(double a, double s, double) { return a / s; }
};

// safe range of the fast sequences (see the header): every nonzero magnitude that enters a fast quotient or root lies in
// [2^-200, 2^200] and every running scale factor in [2^-140, 2^140]: exponent differences stay below 768, no operand or result
// is subnormal, no numerator has a biased exponent <= 53 — the conditions under which v_div_scale / v_div_fixup are the identity
constexpr double RANGE_LO = 0x1p-200, RANGE_HI = 0x1p200, SCALE_LO = 0x1p-140, SCALE_HI = 0x1p140;

}  // namespace acc

}  // namespace wlsqm
