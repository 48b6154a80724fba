// wlsqm_kernels.hpp — per-case WLSQM arithmetic shared by the HIP kernels (gfx950 only).
//
// What is computed follows the reference (file:line in /root/reference):
//   c[k,a] scaled monomials   impl.pyx:70-544 (make_c_{3,2,1}D), DOF order defs.pyx:91-183
//   w[k]                      infra.pyx:668-702 (Case_make_weights, alpha = 1e-4)
//   M = C^T W C, g = C^T W f  impl.pyx:566-602 (make_A), impl.pyx:768-787 (RHS)
//   knowns elimination        impl.pyx:792-823
// How it is computed is MI355X-native and deliberately different: the full `no x no`
// symmetric normal matrix lives in registers (upper triangle), knowns are eliminated by
// masking rows/columns to identity (arithmetically the reduced system of infra.remap,
// with no index indirection), and the system is solved by an unpivoted LDL^T — Cholesky
// is invariant to diagonal scaling, so the reference's Ruiz equilibration
// (lapackdrivers.pyx:553-623) + partial-pivot LU (dgetrf) is not needed for accuracy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wlsqm {

__host__ __device__ constexpr int ndofs(int dim, int order) {
    // defs.pyx:97-101, 127-131, 177-181
    return dim == 1 ? order + 1
         : dim == 2 ? (order + 1) * (order + 2) / 2
                    : (order + 1) * (order + 2) * (order + 3) / 6;
}

// Exponents (p,q,r) of DOF index a: c[a] = dx^p dy^q dz^r / (p! q! r!)
template <int DIM> struct Mono;
template <> struct Mono<1> {
    static constexpr int P[5] = {0, 1, 2, 3, 4};
    static constexpr int Q[5] = {0, 0, 0, 0, 0};
    static constexpr int R[5] = {0, 0, 0, 0, 0};
};
template <> struct Mono<2> {  // defs.pyx:107-125
    static constexpr int P[15] = {0, 1, 0, 2, 1, 0, 3, 2, 1, 0, 4, 3, 2, 1, 0};
    static constexpr int Q[15] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};
    static constexpr int R[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};
template <> struct Mono<3> {  // defs.pyx:137-171 (note the cyclic X2,XY,Y2,YZ,Z2,XZ order)
    static constexpr int P[35] = {0, 1, 0, 0, 2, 1, 0, 0, 0, 1, 3, 2, 1, 0, 0, 0, 0, 1, 2, 1,
                                  4, 3, 2, 1, 0, 0, 0, 0, 0, 1, 2, 3, 2, 1, 1};
    static constexpr int Q[35] = {0, 0, 1, 0, 0, 1, 2, 1, 0, 0, 0, 1, 2, 3, 2, 1, 0, 0, 0, 1,
                                  0, 1, 2, 3, 4, 3, 2, 1, 0, 0, 0, 0, 1, 2, 1};
    static constexpr int R[35] = {0, 0, 0, 1, 0, 0, 0, 1, 2, 1, 0, 0, 0, 0, 1, 2, 3, 2, 1, 1,
                                  0, 0, 0, 0, 0, 1, 2, 3, 4, 3, 2, 1, 1, 1, 2};
};

// index of (a,b), a <= b, in the packed upper triangle of an N x N symmetric matrix
template <int N> __host__ __device__ constexpr int tri(int a, int b) { return a * N - a * (a - 1) / 2 + (b - a); }
template <int N> __host__ __device__ constexpr int sym(int a, int b) { return a <= b ? tri<N>(a, b) : tri<N>(b, a); }

// Scaled powers s[i] = d^i / i!, with the reference's grouping (impl.pyx:319-349):
// d2 = d*d, d3 = d2*d, s2 = 0.5*d2, s3 = (1/6)*d3, s4 = (1/24)*d2*d2.
template <int ORDER> __device__ __forceinline__ void scaled_powers(double d, double (&s)[5]) {
    s[0] = 1.0; s[1] = d;
    const double d2 = d * d;
    s[2] = 0.5 * d2;
    s[3] = (1.0 / 6.0) * (d2 * d);
    s[4] = ((1.0 / 24.0) * d2) * d2;
}

// c[a] for one neighbour offset d[DIM]; returns squared distance.
template <int DIM, int ORDER>
__device__ __forceinline__ double monomials(const double (&d)[DIM], double (&c)[ndofs(DIM, ORDER)]) {
    constexpr int NO = ndofs(DIM, ORDER);
    double sx[5], sy[5], sz[5];
    scaled_powers<ORDER>(d[0], sx);
    double d2 = d[0] * d[0];
    if constexpr (DIM >= 2) { scaled_powers<ORDER>(d[1], sy); d2 += d[1] * d[1]; }
    if constexpr (DIM == 3) { scaled_powers<ORDER>(d[2], sz); d2 += d[2] * d[2]; }
#pragma unroll
    for (int a = 0; a < NO; ++a) {
        double v = sx[Mono<DIM>::P[a]];
        if constexpr (DIM >= 2) { if (Mono<DIM>::Q[a] > 0) v = (Mono<DIM>::P[a] > 0) ? v * sy[Mono<DIM>::Q[a]] : sy[Mono<DIM>::Q[a]]; }
        if constexpr (DIM == 3) { if (Mono<DIM>::R[a] > 0) v = (Mono<DIM>::P[a] + Mono<DIM>::Q[a] > 0) ? v * sz[Mono<DIM>::R[a]] : sz[Mono<DIM>::R[a]]; }
        c[a] = v;
    }
    return d2;
}

// sqrt(q) for q in [0, 1]: v_rsq_f64 seed + two Goldschmidt/Newton steps (<= 1 ulp).  hipcc's generic
// sqrt() spends another ~10 instructions on range scaling for denormal/huge arguments that cannot occur for
// a ratio of squared distances; q == 0 (a neighbour coincident with xi) is handled explicitly.
__device__ __forceinline__ double sqrt_unit(double q) {
    const double y = __builtin_amdgcn_rsq(q);
    double g = q * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, g, q);
    g = fma(r, h, g);
    r = fma(-g, g, q);
    g = fma(r, h, g);
    return q > 0.0 ? g : q;          // q == 0 -> 0; NaN (max_d2 == 0, infra.pyx:701) propagates
}

// infra.pyx:668-702.  UNIFORM: 1; otherwise (CENTER and any other value, :691):
// alpha + beta*(1 - sqrt(d2/max_d2))^2, alpha = 1e-4.  `inv_max` = 1/max_d2 is formed once per case
// (one IEEE divide) and multiplied in: the quotient differs from d2/max_d2 by at most 1 ulp.
__device__ __forceinline__ double weight(double d2, double inv_max, bool uniform) {
    const double t = 1.0 - sqrt_unit(d2 * inv_max);
    const double w = 1e-4 + (1.0 - 1e-4) * t * t;
    return uniform ? 1.0 : w;        // select, not a branch: keeps the neighbour loop straight-line
}
// 1/x for the pivots and 1/max_d2: v_rcp_f64 seed + two Newton steps (full double accuracy for normal
// x; 0 -> inf and NaN propagate like IEEE division).  hipcc's IEEE a/b is a 12-instruction sequence.
__device__ __forceinline__ double recip(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    const double z = fma(y, e, y);
    return (z == z) ? z : y;         // x = 0 or inf: the Newton step makes inf*0 = NaN, keep the raw seed (inf / 0)
}
__device__ __forceinline__ double inverse_max(double max_d2) { return recip(max_d2); }

// Effective knowns mask over the `NO` DOFs.  The reference sizes the reduced system as
// nr = no - popcountll(mask) WITHOUT masking bits >= no (infra.pyx:119-121) but builds
// r2o from bits < no only (infra.pyx:178-198); with stray high bits the last (k - nr)
// unknowns therefore drop out of the system (neither solved nor written).  `dropped`
// reproduces that: those DOFs are eliminated with value 0 and never written.
template <int NO>
__device__ __forceinline__ void effective_mask(long long raw, unsigned long long& known, unsigned long long& dropped) {
    constexpr unsigned long long FULL = (NO >= 64) ? ~0ull : ((1ull << NO) - 1ull);
    known = (unsigned long long)raw & FULL;
    dropped = 0;
    const unsigned long long high = (unsigned long long)raw & ~FULL;
    if (high) {
        int extra = __popcll(high);
        for (int t = NO - 1; t >= 0 && extra > 0; --t)
            if (!((known >> t) & 1ull)) { known |= 1ull << t; dropped |= 1ull << t; --extra; }
    }
}

// In-register LDL^T of the packed upper triangle (M = U^T D U, unit U stored above the
// diagonal, D on it).  Replaces lapackdrivers.pyx:1628-1635 (dgetrf) for this SPD system.
// The updates are written as explicit fma() (here, in ldlt_solve and in eliminate_knowns): `a -= b * c` leaves the compiler a
// choice when `a` is itself a product — M right after the expansion from the moments is (moment x constant) —, namely which of
// the two products to fuse, and two inlined copies of the same routine were seen to choose differently (round 3: a branch-free
// copy of the 3D ring kernel's solve differed from the branching one by 1e-8 relative in 75 % of the cases).  With the fused
// operation spelled out every copy rounds the same way, so a case's bits do not depend on the kernel variant that solved it.
// Up to 10 unknowns only: for the 14- / 15-unknown systems the compiler's freedom is what keeps the 2D order-4 ring kernel's
// matrix out of scratch (it fuses the expansion's products straight into the first updates instead of materialising all 105
// entries: with explicit fma() there the kernel spilled on its hot path, C3 0.46 -> 0.555 ms); those kernels have one copy of
// the solve each.
__host__ __device__ constexpr bool pin_fma(int n) { return n <= 10; }
template <int N> __device__ __forceinline__ void ldlt_factor(double (&M)[N * (N + 1) / 2]) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double inv = recip(M[tri<N>(j, j)]);
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            const double t = M[tri<N>(j, i)] * inv;
#pragma unroll
            for (int m = i; m < N; ++m) {
                if constexpr (pin_fma(N)) M[tri<N>(i, m)] = fma(-t, M[tri<N>(j, m)], M[tri<N>(i, m)]);
                else M[tri<N>(i, m)] -= t * M[tri<N>(j, m)];
            }
            M[tri<N>(j, i)] = t;
        }
        M[tri<N>(j, j)] = inv;   // keep 1/d_j: the solves only ever divide by d_j
    }
}

// Solve with the factor from ldlt_factor; b is overwritten with x.  Replaces dgetrs
// (lapackdrivers.pyx:1657-1665).
template <int N> __device__ __forceinline__ void ldlt_solve(const double (&M)[N * (N + 1) / 2], double (&b)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            if constexpr (pin_fma(N)) b[i] = fma(-M[tri<N>(j, i)], b[j], b[i]);
            else b[i] -= M[tri<N>(j, i)] * b[j];
        }
#pragma unroll
    for (int j = N - 1; j >= 0; --j) {
        double v = b[j] * M[tri<N>(j, j)];
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            if constexpr (pin_fma(N)) v = fma(-M[tri<N>(j, i)], b[i], v);
            else v -= M[tri<N>(j, i)] * b[i];
        }
        b[j] = v;
    }
}

// Knowns elimination (impl.pyx:792-818) + masking to identity.  `val[a]` is fi[a] for true
// knowns and 0 for dropped DOFs.
template <int N>
__device__ __forceinline__ void eliminate_knowns(double (&M)[N * (N + 1) / 2], double (&g)[N],
                                                 unsigned long long known, const double (&val)[N]) {
#pragma unroll
    for (int om = 0; om < N; ++om) {
        if ((known >> om) & 1ull) {
            const double v = val[om];
#pragma unroll
            for (int a = 0; a < N; ++a)
                if (a != om) {
                    if constexpr (pin_fma(N)) g[a] = fma(-M[sym<N>(a, om)], v, g[a]);
                    else g[a] -= M[sym<N>(a, om)] * v;
                }
        }
    }
#pragma unroll
    for (int om = 0; om < N; ++om) {
        if ((known >> om) & 1ull) {
#pragma unroll
            for (int a = 0; a < N; ++a)
                if (a != om) M[sym<N>(a, om)] = 0.0;
            M[tri<N>(om, om)] = 1.0;
            g[om] = 0.0;
        }
    }
}

// Accumulate one neighbour into (M, g): impl.pyx:601 and :774/:786.
template <int N>
__device__ __forceinline__ void accumulate(double (&M)[N * (N + 1) / 2], double (&g)[N],
                                           const double (&c)[N], double w, double f) {
    const double wf = w * f;
    double t[N];
#pragma unroll
    for (int a = 0; a < N; ++a) t[a] = (a == 0) ? w : w * c[a];   // c[0] == 1
#pragma unroll
    for (int a = 0; a < N; ++a) g[a] += (a == 0) ? wf : wf * c[a];
#pragma unroll
    for (int b = 0; b < N; ++b) M[tri<N>(0, b)] += t[b];           // row 0: (w c_b) * 1
#pragma unroll
    for (int a = 1; a < N; ++a)
#pragma unroll
        for (int b = a; b < N; ++b) M[tri<N>(a, b)] += t[a] * c[b];
}

}  // namespace wlsqm
