// wlsqm_moments.hpp — moment form of the normal equations.
//
//   M[a,b] = sum_k w_k c_a(k) c_b(k) = mu(P_a + P_b) / (P_a! P_b!)      mu(P) = sum_k w_k d^P
//   g[a]   = sum_k w_k f_k c_a(k)    = nu(P_a) / P_a!                   nu(P) = sum_k w_k f_k d^P
// with c_a = d^{P_a} / P_a! (impl.pyx:102-157, 319-349, 476-489; M as impl.pyx:601, g as :774).  The
// no(no+1)/2 matrix entries are only C(2*order + dim, dim) distinct moments (2D order 2: 15 of 21,
// 3D order 2: 35 of 55, 2D order 4: 45 of 120), accumulated with one multiply per monomial
// (t(P) = t(P - e_m) d_m) and one add; the top degree is folded into an FMA.
#pragma once
#include "wlsqm_kernels.hpp"

namespace wlsqm {

__host__ __device__ constexpr int mtri(int d) { return d * (d + 1) / 2; }
__host__ __device__ constexpr int mtet(int d) { return d * (d + 1) * (d + 2) / 6; }
// number of monomials of total degree <= deg
template <int DIM> __host__ __device__ constexpr int mom_count(int deg) {
    return DIM == 1 ? deg + 1 : DIM == 2 ? mtri(deg + 1) : mtet(deg + 1);
}
// graded index of x^p y^q z^r: degrees ascending; inside a degree by (q + r) ascending, then r ascending
template <int DIM> __host__ __device__ constexpr int mom_index(int p, int q, int r) {
    return DIM == 1 ? p : DIM == 2 ? mtri(p + q) + q : mtet(p + q + r) + mtri(q + r) + r;
}
__host__ __device__ constexpr double mom_inv_fact(int n) {
    return n <= 1 ? 1.0 : n == 2 ? 0.5 : n == 3 ? 1.0 / 6.0 : 1.0 / 24.0;
}

// One neighbour into the accumulators: mu over degrees 0..2*ORDER, nu over degrees 0..ORDER (both in graded order).
template <int DIM, int ORDER>
__device__ __forceinline__ void accumulate_moments(double (&mu)[mom_count<DIM>(2 * ORDER)], double (&nu)[mom_count<DIM>(ORDER)],
                                                   const double (&d)[DIM], double w, double f) {
    constexpr int D = 2 * ORDER;
    constexpr int W = DIM == 1 ? 1 : DIM == 2 ? D + 1 : mtri(D + 1);   // monomials in the widest degree kept
    double prev[W], cur[W];
    prev[0] = w;
    mu[0] += w;
    nu[0] = fma(w, f, nu[0]);
#pragma unroll
    for (int deg = 1; deg <= D; ++deg) {
        const int base = mom_count<DIM>(deg - 1);                      // index of the first monomial of this degree
        // enumerate the monomials of this degree in graded order; `pos` is the position inside the degree
        const int smax = (DIM == 1) ? 0 : deg;
#pragma unroll
        for (int s = 0; s <= smax; ++s) {                              // s = q + r  (DIM == 2: s = q)
            const int rmax = (DIM == 3) ? s : 0;
#pragma unroll
            for (int r = 0; r <= rmax; ++r) {
                const int q = s - r;
                const int pos = (DIM == 3) ? mtri(s) + r : s;
                // parent monomial (degree deg-1) and the variable that takes it here
                int ppos; int var;
                if (DIM == 3 && r > 0) { ppos = mtri(s - 1) + (r - 1); var = 2; }          // (p, q, r-1) * z
                else if (DIM >= 2 && q > 0) { ppos = (DIM == 3) ? mtri(s - 1) : s - 1; var = 1; }   // (p, q-1, 0) * y
                else { ppos = (DIM == 3) ? mtri(s) + r : s; var = 0; }                      // (p-1, q, r) * x
                // for var == 0 the parent has the same (q, r), i.e. the same position in the previous degree
                if (deg < D) {
                    const double t = prev[ppos] * d[var];
                    cur[pos] = t;
                    mu[base + pos] += t;
                    if (deg <= ORDER) nu[base + pos] = fma(t, f, nu[base + pos]);
                } else {
                    mu[base + pos] = fma(prev[ppos], d[var], mu[base + pos]);
                }
            }
        }
        if (deg < D) {
            const int cnt = mom_count<DIM>(deg) - base;
#pragma unroll
            for (int i = 0; i < W; ++i) if (i < cnt) prev[i] = cur[i];
        }
    }
}

// 2D alternative: outer-product form.  With X[p] = dx^p and Yw[q] = w dy^q every moment is ONE fma,
// mu(p,q) += X[p] * Yw[q], and nu(p,q) += X[p] * (f Yw[q]); the chain form above spends a multiply AND an add on every
// monomial below the top degree.  Operations per neighbour, order 4: 45 + 15 fma + 21 multiplies = 81 against 96;
// order 2: 31 against 30 (no gain: used from order 3 up, see accumulate_moments_best).  The 3D analogue (XY[p][q] * Zw[r])
// has the chain form's operation count (64 for order 2) and measured the same on C5 (+-1 %): not kept.
template <int ORDER>
__device__ __forceinline__ void accumulate_moments_outer2d(double (&mu)[mom_count<2>(2 * ORDER)], double (&nu)[mom_count<2>(ORDER)],
                                                           const double (&d)[2], double w, double f) {
    constexpr int D = 2 * ORDER;
    double X[D + 1], Yw[D + 1], Yf[ORDER + 1];
    X[1] = d[0];
#pragma unroll
    for (int p = 2; p <= D; ++p) X[p] = X[p - 1] * d[0];
    Yw[0] = w;
#pragma unroll
    for (int q = 1; q <= D; ++q) Yw[q] = Yw[q - 1] * d[1];
#pragma unroll
    for (int q = 0; q <= ORDER; ++q) Yf[q] = Yw[q] * f;
#pragma unroll
    for (int deg = 0; deg <= D; ++deg) {
#pragma unroll
        for (int q = 0; q <= deg; ++q) {
            const int p = deg - q, i = mtri(deg) + q;
            if (p == 0) { mu[i] += Yw[q]; if (deg <= ORDER) nu[i] += Yf[q]; }
            else { mu[i] = fma(X[p], Yw[q], mu[i]); if (deg <= ORDER) nu[i] = fma(X[p], Yf[q], nu[i]); }
        }
    }
}

// 3D, order 3 and up: powers of x and y once per neighbour, then for every (q, r) the product yz = (w z^r) y^q and ONE fma per moment,
// mu(p,q,r) += x^p yz.  Order 3: 37 multiplies + 10 for the right-hand side + 104 accumulations = 151 operations and 14 live
// temporaries, against 203 instructions (25 of them copies) and 56 temporaries of the chain form — with the 104 accumulators of the
// 20-unknown systems (208 of the 256 registers an instruction can name) the temporaries decide whether accumulators travel through
// the accumulation file inside the neighbour loop.
// PART = 1 / 2: only the moments whose z power r has stage_part(r) == PART (the two halves of the 165 + 35 sums of 3D order 4, 104 and
// 96 accumulators: fit_stage_kernel<3,4,PART>).
__host__ __device__ constexpr int stage_part(int r) { return (r == 0 || r == 2 || r >= 6) ? 1 : 2; }
template <int ORDER, int PART = 0>
__device__ __forceinline__ void accumulate_moments_outer3d(double (&mu)[mom_count<3>(2 * ORDER)], double (&nu)[mom_count<3>(ORDER)],
                                                           const double (&d)[3], double w, double f) {
    constexpr int D = 2 * ORDER;
    double X[D + 1], Y[D + 1];
    X[1] = d[0]; Y[1] = d[1];
#pragma unroll
    for (int p = 2; p <= D; ++p) { X[p] = X[p - 1] * d[0]; Y[p] = Y[p - 1] * d[1]; }
    double zw = w;
#pragma unroll
    for (int r = 0; r <= D; ++r) {
        if (r > 0) zw *= d[2];
        if (PART != 0 && stage_part(r) != PART) continue;
#pragma unroll
        for (int q = 0; q + r <= D; ++q) {
            const double yz = q > 0 ? zw * Y[q] : zw;
            mu[mom_index<3>(0, q, r)] += yz;
#pragma unroll
            for (int p = 1; p + q + r <= D; ++p) mu[mom_index<3>(p, q, r)] = fma(X[p], yz, mu[mom_index<3>(p, q, r)]);
            if (q + r <= ORDER) {
                const double yzf = yz * f;
                nu[mom_index<3>(0, q, r)] += yzf;
#pragma unroll
                for (int p = 1; p + q + r <= ORDER; ++p) nu[mom_index<3>(p, q, r)] = fma(X[p], yzf, nu[mom_index<3>(p, q, r)]);
            }
        }
    }
}

// ONE pass for neighbours in ANY order (round 5).  The weight needs the largest squared distance of the case, which unsorted input does
// not reveal before the last neighbour; but alpha + beta (1 - sqrt(d2 / max))^2 = (alpha + beta) - 2 beta sqrt(d2) / sqrt(max) + beta d2 / max
// is LINEAR in 1, sqrt(d2) and d2 with coefficients that depend on the maximum alone: three sets of moments — sum P, sum sqrt(d2) P,
// sum d2 P over the unweighted monomials P — are summed in one pass and combined at the end (combine_triple).  Three accumulators per
// moment (2D order 2: 63 instead of 21) and three fused multiply-adds; the combination cancels a few bits (the three terms are of the
// size of sum P, the result of the size of the mean weight, 0.2-0.3, times that).  Used where the accumulators fit (2D up to order 2).
template <int DIM, int ORDER>
__device__ __forceinline__ void accumulate_moments_triple(double (&mu3)[3][mom_count<DIM>(2 * ORDER)], double (&nu3)[3][mom_count<DIM>(ORDER)],
                                                          const double (&d)[DIM], double one, double r, double d2, double f) {
    constexpr int D = 2 * ORDER, NM = mom_count<DIM>(D);
    double P[NM];
    P[0] = 1.0;
    mu3[0][0] += one; mu3[1][0] += r; mu3[2][0] += d2;                 // (one = 0 for a masked slot, whose d, r, d2 and f are 0 too)
    nu3[0][0] += f; nu3[1][0] = fma(f, r, nu3[1][0]); nu3[2][0] = fma(f, d2, nu3[2][0]);
#pragma unroll
    for (int deg = 1; deg <= D; ++deg) {
        const int base = mom_count<DIM>(deg - 1), pbase = deg >= 2 ? mom_count<DIM>(deg - 2) : 0;
        const int smax = (DIM == 1) ? 0 : deg;
#pragma unroll
        for (int s = 0; s <= smax; ++s) {
            const int rmax = (DIM == 3) ? s : 0;
#pragma unroll
            for (int rr = 0; rr <= rmax; ++rr) {
                const int q = s - rr;
                const int pos = (DIM == 3) ? mtri(s) + rr : s;
                int ppos; int var;                                     // parent monomial (degree deg - 1), as in accumulate_moments
                if (DIM == 3 && rr > 0) { ppos = mtri(s - 1) + (rr - 1); var = 2; }
                else if (DIM >= 2 && q > 0) { ppos = (DIM == 3) ? mtri(s - 1) : s - 1; var = 1; }
                else { ppos = (DIM == 3) ? mtri(s) + rr : s; var = 0; }
                const double t = P[pbase + ppos] * d[var];
                P[base + pos] = t;
                // (the first set exactly as accumulate_moments sums it with the weight 1: a uniformly weighted case gets the same bits here)
                if (deg < D) mu3[0][base + pos] += t; else mu3[0][base + pos] = fma(P[pbase + ppos], d[var], mu3[0][base + pos]);
                mu3[1][base + pos] = fma(t, r, mu3[1][base + pos]);
                mu3[2][base + pos] = fma(t, d2, mu3[2][base + pos]);
                if (deg <= ORDER) {
                    const double tf = t * f;
                    nu3[0][base + pos] = fma(t, f, nu3[0][base + pos]);
                    nu3[1][base + pos] = fma(tf, r, nu3[1][base + pos]);
                    nu3[2][base + pos] = fma(tf, d2, nu3[2][base + pos]);
                }
            }
        }
    }
}
// the weighted moments from the three sets: inv_max = 1 / max_d2, root_inv = sqrt(inv_max); uniform weighting keeps the first set
template <int N>
__device__ __forceinline__ void combine_triple(double (&out)[N], double (&in3)[3][N], double inv_max, double root_inv, bool uniform) {
    const double a1 = -2.0 * (1.0 - 1e-4) * root_inv, a2 = (1.0 - 1e-4) * inv_max;      // (alpha + beta = 1)
#pragma unroll
    for (int i = 0; i < N; ++i) { const double v = fma(a2, in3[2][i], fma(a1, in3[1][i], in3[0][i])); out[i] = uniform ? in3[0][i] : v; }
}

// The cheaper of the two forms for (DIM, ORDER) (the 3D outer form is the staged kernel's: csrc/fit_stage.hip).
template <int DIM, int ORDER>
__device__ __forceinline__ void accumulate_moments_best(double (&mu)[mom_count<DIM>(2 * ORDER)], double (&nu)[mom_count<DIM>(ORDER)],
                                                        const double (&d)[DIM], double w, double f) {
#ifndef WLSQM_NO_OUTER_MOMENTS
    if constexpr (DIM == 2 && ORDER >= 3) { accumulate_moments_outer2d<ORDER>(mu, nu, d, w, f); return; }
#endif
    accumulate_moments<DIM, ORDER>(mu, nu, d, w, f);
}

// Expand the packed upper triangle of M and the right-hand side g (DOF order) from the moments.
// `mu_at(i)` / `nu_at(i)` deliver moment i (graded index); each is asked for exactly once, in ascending order,
// so a caller may fetch (and sum) them from LDS on demand and never hold the whole moment vector in registers.
template <int DIM, int ORDER, class MuAt, class NuAt>
__device__ __forceinline__ void expand_moments_from(MuAt mu_at, NuAt nu_at,
                                                    double (&M)[ndofs(DIM, ORDER) * (ndofs(DIM, ORDER) + 1) / 2],
                                                    double (&g)[ndofs(DIM, ORDER)]) {
    constexpr int NO = ndofs(DIM, ORDER), NM = mom_count<DIM>(2 * ORDER);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const double m = mu_at(i);
#pragma unroll
        for (int a = 0; a < NO; ++a) {
            const int pa = Mono<DIM>::P[a], qa = Mono<DIM>::Q[a], ra = Mono<DIM>::R[a];
            const double fa = mom_inv_fact(pa) * mom_inv_fact(qa) * mom_inv_fact(ra);
#pragma unroll
            for (int b = a; b < NO; ++b) {
                const int pb = Mono<DIM>::P[b], qb = Mono<DIM>::Q[b], rb = Mono<DIM>::R[b];
                const double fb = mom_inv_fact(pb) * mom_inv_fact(qb) * mom_inv_fact(rb);
                if (mom_index<DIM>(pa + pb, qa + qb, ra + rb) == i) M[tri<NO>(a, b)] = m * (fa * fb);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < NO; ++a) {
        const int pa = Mono<DIM>::P[a], qa = Mono<DIM>::Q[a], ra = Mono<DIM>::R[a];
        g[a] = nu_at(mom_index<DIM>(pa, qa, ra)) * (mom_inv_fact(pa) * mom_inv_fact(qa) * mom_inv_fact(ra));
    }
}

template <int DIM, int ORDER>
__device__ __forceinline__ void expand_moments(const double (&mu)[mom_count<DIM>(2 * ORDER)],
                                               const double (&nu)[mom_count<DIM>(ORDER)],
                                               double (&M)[ndofs(DIM, ORDER) * (ndofs(DIM, ORDER) + 1) / 2],
                                               double (&g)[ndofs(DIM, ORDER)]) {
    expand_moments_from<DIM, ORDER>([&](int i) { return mu[i]; }, [&](int i) { return nu[i]; }, M, g);
}

}  // namespace wlsqm
