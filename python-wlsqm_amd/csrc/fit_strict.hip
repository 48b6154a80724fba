// fit_strict.hip — the REFERENCE-ORDER numerics mode (WLSQM_HIP_STRICT=1 / wlsqm_hip_set_strict(1)) for gfx950.
//
// The fast kernels (fit_tile / fit_ring / ...) compute the same fit with MI355X-first arithmetic: moment form, neighbour sums
// split over lanes, rsq / rcp seeded weights, FMA contraction, unpivoted LDL^T.  Their result differs from the reference's by
// kappa * eps-type rounding, which at the density the metric is quoted on (1M points: h ~ 0.003) exceeds 1e-10 on the second
// derivatives.  This kernel instead replays the reference's floating-point operations ONE FOR ONE, so that its output differs
// from the reference's only where LAPACK's internal summation order differs from the textbook algorithm:
//
//   make_c_{1,2,3}D      impl.pyx:449-544, 286-432, 70-269   same grouping of every scaled monomial
//   Case_make_weights    infra.pyx:668-702                    IEEE divide and IEEE sqrt per neighbour
//   make_A               impl.pyx:566-602                     ALL nr^2 entries, each summed over k ascending in ONE lane,
//                                                             term (w[k] * c[k,om]) * c[k,oj], no FMA contraction
//   rescale_ruiz2001_c   lapackdrivers.pyx:553-623            iterative equilibration, same stop test, <= 100 sweeps
//   apply_scaling_c      lapackdrivers.pyx:293-299
//   dgetrf               lapackdrivers.pyx:1628-1635          unblocked partial-pivot LU (first maximum wins, reciprocal pivot)
//   solve / solve_contig impl.pyx:731-974                     RHS, knowns elimination term by term, dgetrs, un-scaling, do_sens
//   solve_iterative      impl.pyx:986-1083                    refinement with the FMA Horner model of polyeval.pyx
//
// Three mappings of the same operations (no sum is ever split in any of them):
//   fit_strict_reg_kernel   one lane per case, everything in registers: basic fits of systems up to 10 unknowns, 64-case groups
//                           without knowns or with exactly F known
//   fit_strict_rows_kernel  one lane per matrix ROW (a case on 2..64 lanes), neighbours in LDS: the larger systems, sensitivities,
//                           refinement
//   fit_strict_kernel       one lane per case with the reduced matrix and every per-case vector in LDS as [slot][lane]: the rest
//                           (mixed groups, neighbourhoods beyond the LDS, capture of the intermediates)
// It is a correctness mode: 5-20x the time of the fast kernels, chosen per call or per process, never silently.
#include <atomic>
#include <type_traits>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_strict.hpp"

#pragma clang fp contract(off)      // the reference is gcc -O2 on x86-64: no contraction; the explicit fma() calls below are polyeval.pyx's own

#ifndef WLSQM_STRICT_ROWS_MINW
#define WLSQM_STRICT_ROWS_MINW 0      // A/B: waves per SIMD the row-per-lane kernel is compiled for (0: rows_minw below)
#endif

namespace wlsqm {

namespace strict {

// polyeval.pyx:874-951 (1D), :550-735 (2D), :82-355 (3D): the model at one neighbour, fi[] read through `f(a)`
template <int DIM, int ORDER, class FI>
__device__ __forceinline__ double taylor(const double (&d)[DIM], const FI& f) {
    if constexpr (ORDER == 0) return f(0);
    if constexpr (DIM == 1) {
        const double dx = d[0];
        double acc;
        if constexpr (ORDER == 4) {
            acc = fma(dx, one24th * f(4), onesixth * f(3)); acc = fma(dx, acc, 0.5 * f(2)); acc = fma(dx, acc, f(1));
            return fma(dx, acc, f(0));
        } else if constexpr (ORDER == 3) {
            acc = fma(dx, onesixth * f(3), 0.5 * f(2)); acc = fma(dx, acc, f(1));
            return fma(dx, acc, f(0));
        } else if constexpr (ORDER == 2) {
            acc = fma(dx, 0.5 * f(2), f(1));
            return fma(dx, acc, f(0));
        } else {
            return fma(dx, f(1), f(0));
        }
    } else if constexpr (DIM == 2) {
        const double dx = d[0], dy = d[1], dxdy = dx * dy;
        double acc1, acc2, resX, resY;
        if constexpr (ORDER == 4) {
            acc1 = fma(dy, f(11), f(6)); acc1 *= onesixth; acc1 = fma(dx, one24th * f(10), acc1);
            acc2 = fma(dy, f(7), f(3)); acc2 *= 0.5; acc2 = fma(dx, acc1, acc2);
            resX = fma(dx, acc2, f(1));
            acc1 = fma(dx, f(13), f(9)); acc1 *= onesixth; acc1 = fma(dy, one24th * f(14), acc1);
            acc2 = fma(dx, f(8), f(5)); acc2 *= 0.5; acc2 = fma(dy, acc1, acc2);
            resY = fma(dy, acc2, f(2));
            const double resXY = fma(dxdy, 0.25 * f(12), f(4));
            acc1 = dxdy * resXY;
        } else if constexpr (ORDER == 3) {
            acc2 = fma(dy, f(7), f(3)); acc2 *= 0.5; acc2 = fma(dx, onesixth * f(6), acc2);
            resX = fma(dx, acc2, f(1));
            acc2 = fma(dx, f(8), f(5)); acc2 *= 0.5; acc2 = fma(dy, onesixth * f(9), acc2);
            resY = fma(dy, acc2, f(2));
            acc1 = dxdy * f(4);
        } else if constexpr (ORDER == 2) {
            resX = fma(dx, 0.5 * f(3), f(1));
            resY = fma(dy, 0.5 * f(5), f(2));
            acc1 = dxdy * f(4);
        } else {
            acc1 = dx * f(1); acc1 = fma(dy, f(2), acc1); acc1 += f(0);
            return acc1;
        }
        acc1 = fma(dx, resX, acc1); acc1 = fma(dy, resY, acc1); acc1 += f(0);
        return acc1;
    } else {
        const double dx = d[0], dy = d[1], dz = d[2];
        const double dxdy = dx * dy, dydz = dy * dz, dxdz = dx * dz;
        double acc1, acc2, resX, resY, resZ;
        if constexpr (ORDER == 4) {
            acc1 = fma(dy, f(21), f(10)); acc1 = fma(dz, f(31), acc1); acc1 *= onesixth; acc1 = fma(dx, one24th * f(20), acc1);
            acc2 = fma(dy, f(11), f(4)); acc2 = fma(dz, f(18), acc2); acc2 = fma(dydz, f(32), acc2); acc2 *= 0.5;
            acc2 = fma(dx, acc1, acc2);
            resX = fma(dx, acc2, f(1));
            acc1 = fma(dx, f(23), f(13)); acc1 = fma(dz, f(25), acc1); acc1 *= onesixth; acc1 = fma(dy, one24th * f(24), acc1);
            acc2 = fma(dx, f(12), f(6)); acc2 = fma(dz, f(14), acc2); acc2 = fma(dxdz, f(33), acc2); acc2 *= 0.5;
            acc2 = fma(dy, acc1, acc2);
            resY = fma(dy, acc2, f(2));
            acc1 = fma(dx, f(29), f(16)); acc1 = fma(dy, f(27), acc1); acc1 *= onesixth; acc1 = fma(dz, one24th * f(28), acc1);
            acc2 = fma(dx, f(17), f(8)); acc2 = fma(dy, f(15), acc2); acc2 = fma(dxdy, f(34), acc2); acc2 *= 0.5;
            acc2 = fma(dz, acc1, acc2);
            resZ = fma(dz, acc2, f(3));
            const double resXY = fma(dxdy, 0.25 * f(22), f(5));
            const double resYZ = fma(dydz, 0.25 * f(26), f(7));
            const double resXZ = fma(dxdz, 0.25 * f(30), f(9));
            acc1 = dx * dy * dz * f(19);
            acc1 = fma(dxdy, resXY, acc1); acc1 = fma(dydz, resYZ, acc1); acc1 = fma(dxdz, resXZ, acc1);
        } else if constexpr (ORDER == 3) {
            acc2 = fma(dy, f(11), f(4)); acc2 = fma(dz, f(18), acc2); acc2 *= 0.5; acc2 = fma(dx, onesixth * f(10), acc2);
            resX = fma(dx, acc2, f(1));
            acc2 = fma(dx, f(12), f(6)); acc2 = fma(dz, f(14), acc2); acc2 *= 0.5; acc2 = fma(dy, onesixth * f(13), acc2);
            resY = fma(dy, acc2, f(2));
            acc2 = fma(dx, f(17), f(8)); acc2 = fma(dy, f(15), acc2); acc2 *= 0.5; acc2 = fma(dz, onesixth * f(16), acc2);
            resZ = fma(dz, acc2, f(3));
            acc1 = dx * dy * dz * f(19);
            acc1 = fma(dxdy, f(5), acc1); acc1 = fma(dydz, f(7), acc1); acc1 = fma(dxdz, f(9), acc1);
        } else if constexpr (ORDER == 2) {
            resX = fma(dx, 0.5 * f(4), f(1));
            resY = fma(dy, 0.5 * f(6), f(2));
            resZ = fma(dz, 0.5 * f(8), f(3));
            acc1 = dxdy * f(5);
            acc1 = fma(dydz, f(7), acc1); acc1 = fma(dxdz, f(9), acc1);
        } else {
            acc1 = dx * f(1); acc1 = fma(dy, f(2), acc1); acc1 = fma(dz, f(3), acc1); acc1 += f(0);
            return acc1;
        }
        acc1 = fma(dx, resX, acc1); acc1 = fma(dy, resY, acc1); acc1 = fma(dz, resZ, acc1); acc1 += f(0);
        return acc1;
    }
}

}  // namespace strict

__device__ __forceinline__ bool fit_strict_group_is_plain(const KParams& p, long long t, long long ncases, long long want = 0);

// (vblock: the workgroup's number in the batch)
template <int DIM, int ORDER>
__device__ __forceinline__ void fit_strict_block(const KParams& p, const StrictDebug& dbg, const int skip_plain_groups, const long long vblock,
                                                 double* smem) {
    using namespace strict;
    constexpr int NO = ndofs(DIM, ORDER);
    constexpr int LPW = lanes_for(NO);
    const int lane = threadIdx.x;
    if (skip_plain_groups == 1) {
        // the register kernels take the 64-case groups without any known DOF and those with exactly F known everywhere: this block's
        // cases lie in group (block * LPW) / 64
        const long long g0 = (vblock * LPW) / 64 * 64;
        if (fit_strict_group_is_plain(p, g0 + lane, live_cases(p), 0)) return;
        if (NO >= 2 && fit_strict_group_is_plain(p, g0 + lane, live_cases(p), 1)) return;      // the F-known register kernel has it
    }
    if (lane >= LPW) return;
    const long long t = vblock * LPW + lane;
    if (t >= live_cases(p)) return;
    const long long j = p.case_index ? p.case_index[t] : t;

    // LDS image of this case: slot s at smem[s * LPW + lane]
    double* const base = smem + lane;
    auto A = [&](int e) -> double& { return base[e * LPW]; };                    // NO*NO, then nr*nr packed at the front
    auto V = [&](int v, int i) -> double& { return base[(NO * NO + v * NO + i) * LPW]; };
    enum { RS = 0, CS, DRP, DCP, DR, DC, BB, R2O, FI, WFI };                     // 10 vectors of NO slots

    const int nk = min(p.nk[j * p.snk], (int)p.max_nk);
    const bool uniform = (p.wm[j * p.swm] == WLSQM_WEIGHT_UNIFORM);
    const long long knowns = p.knowns[j * p.sknowns];
    const int nr = NO - __popcll((unsigned long long)knowns);                    // infra.pyx:119-121: bits >= no are not masked
    if (nr < 1) {                                                                // impl.pyx:574, 636, 742: every step is a no-op ...
        // ... except the refinement loop itself (impl.pyx:1026-1081), which still runs: fi never changes, so the second pass finds
        // the residual norm of the first and stops (count 1) — unless the norm is NaN, which equals nothing: then all max_iter
        // passes run.  norm starts as |res[0]| and `tmp > norm` is false for NaN on either side: it is NaN iff res[0] is.
        if (p.iterative && p.iters_out) {
            int iters = 1;
            if (nk > 0 && p.max_iter > 1) {
                double xi0[DIM], d0[DIM];
                const double* fio0 = p.fi + j * p.sfi_j;
                Rows<DIM> r0;
                if (p.hoods) {
                    const long long pj = own_point(p, j);
#pragma unroll
                    for (int m = 0; m < DIM; ++m) xi0[m] = p.S[pj * DIM + m];
                    r0 = Rows<DIM>{nullptr, 0, nullptr, 0, p.hoods + j * p.shoods_j, p.S, p.F};
                } else {
#pragma unroll
                    for (int m = 0; m < DIM; ++m) xi0[m] = p.xi[j * p.sxi_j + m];
                    r0 = Rows<DIM>{p.xk + j * p.sxk_j, p.sxk_k, p.fk + j * p.sfk_j, p.sfk_k, nullptr, nullptr, nullptr};
                }
                r0.offset(0, xi0, d0);
                const double res0 = r0.value(0) - taylor<DIM, ORDER>(d0, [&](int a) { return fio0[a]; });
                if (res0 != res0) iters = p.max_iter;
            }
            atomicMax(p.iters_out, iters);
        }
        return;
    }
    {   // infra.pyx:145-200 (remap): r2o of the unknowns in ascending DOF order
        int k = 0;
        for (int a = 0; a < NO; ++a)
            if (!((knowns >> a) & 1ll)) { V(R2O, k) = (double)a; ++k; }
    }

    double xi[DIM];
    Rows<DIM> rows;
    if (p.hoods) {
        const long long pj = own_point(p, j);
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.S[pj * DIM + m];
        rows = Rows<DIM>{nullptr, 0, nullptr, 0, p.hoods + j * p.shoods_j, p.S, p.F};
    } else {
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.xi[j * p.sxi_j + m];
        rows = Rows<DIM>{p.xk + j * p.sxk_j, p.sxk_k, p.fk + j * p.sfk_j, p.sfk_k, nullptr, nullptr, nullptr};
    }
    double* const fio = p.fi + j * p.sfi_j;

    // ---- make_c pass 1: largest squared distance
    double max_d2 = 0.;
    for (int k = 0; k < nk; ++k) {
        double d[DIM], c[NO];
        rows.offset(k, xi, d);
        const double d2 = make_c<DIM, ORDER>(d, c);
        if (d2 > max_d2) max_d2 = d2;
    }

    // ---- make_A (impl.pyx:566-602) on the FULL index set: entry (oj, om) = sum_k (w c_om) c_oj, k ascending; the reduced matrix
    // is the sub-array picked by r2o (each entry is its own sum, so computing the unused ones changes nothing)
    for (int e = 0; e < NO * NO; ++e) A(e) = 0.;
    for (int k = 0; k < nk; ++k) {
        double d[DIM], c[NO];
        rows.offset(k, xi, d);
        const double d2 = make_c<DIM, ORDER>(d, c);
        const double w = make_weight(d2, max_d2, uniform);
        if (dbg.w) dbg.w[j * dbg.w_stride + k] = w;
#pragma unroll
        for (int om = 0; om < NO; ++om) {
            const double wc = w * c[om];
#pragma unroll
            for (int oj = 0; oj < NO; ++oj) A(oj + NO * om) += wc * c[oj];
        }
    }
    // compact to A[jj + nr * m] = full[r2o[jj] + NO * r2o[m]]: targets ascend and never pass their sources
    if (nr < NO) {
        for (int m = 0; m < nr; ++m) {
            const int om = (int)V(R2O, m);
            for (int jj = 0; jj < nr; ++jj) A(jj + nr * m) = A((int)V(R2O, jj) + NO * om);
        }
    }
    if (dbg.A)
        for (int e = 0; e < nr * nr; ++e) dbg.A[j * dbg.mat_stride + e] = A(e);

    // ---- rescale_ruiz2001_c (lapackdrivers.pyx:553-623) with init_scaling_c (:285-290).  The row pass divides A[j,m] by
    // DRprev[j] * DCprev[m] and the column pass by DCprev[m] * DRprev[j]: the same double, so one quotient serves both maxima.
    for (int i = 0; i < nr; ++i) { V(RS, i) = 1.; V(CS, i) = 1.; V(DRP, i) = 1.; V(DCP, i) = 1.; }
    for (int it = 0; it < 100; ++it) {
        for (int i = 0; i < nr; ++i) { V(DR, i) = 0.; V(DC, i) = 0.; }
        for (int m = 0; m < nr; ++m) {
            const double cc = V(DCP, m);
            double cmax = 0.;
            for (int jj = 0; jj < nr; ++jj) {
                const double q = fabs(A(jj + nr * m) / (V(DRP, jj) * cc));
                if (q > cmax) cmax = q;
                if (q > V(DR, jj)) V(DR, jj) = q;
            }
            V(DC, m) = sqrt(cmax);
        }
        for (int i = 0; i < nr; ++i) V(DR, i) = sqrt(V(DR, i));
        for (int i = 0; i < nr; ++i) { V(DRP, i) *= V(DR, i); V(RS, i) /= V(DR, i); }
        for (int i = 0; i < nr; ++i) { V(DCP, i) *= V(DC, i); V(CS, i) /= V(DC, i); }
        double acc = fabs(1. - V(DR, 0) * V(DR, 0));
        for (int i = 1; i < nr; ++i) { const double tmp = fabs(1. - V(DR, i) * V(DR, i)); if (tmp > acc) acc = tmp; }
        if (acc < ruiz_epsilon) {
            acc = fabs(1. - V(DC, 0) * V(DC, 0));
            for (int i = 1; i < nr; ++i) { const double tmp = fabs(1. - V(DC, i) * V(DC, i)); if (tmp > acc) acc = tmp; }
            if (acc < ruiz_epsilon) break;
        }
    }
    // apply_scaling_c (lapackdrivers.pyx:293-299)
    for (int m = 0; m < nr; ++m) {
        const double cc = V(CS, m);
        for (int jj = 0; jj < nr; ++jj) A(jj + nr * m) *= (V(RS, jj) * cc);
    }

    // ---- dgetrf (lapackdrivers.pyx:1628-1635; unblocked dgetf2: first maximal |a_ik|, column scaled by the reciprocal pivot).
    // ipiv lives in the DR vector from here on (the equilibration is over).
    for (int c0 = 0; c0 < nr; ++c0) {
        int pv = c0; double best = fabs(A(c0 + nr * c0));
        for (int i = c0 + 1; i < nr; ++i) { const double v = fabs(A(i + nr * c0)); if (v > best) { best = v; pv = i; } }
        V(DR, c0) = (double)(pv + 1);
        if (A(pv + nr * c0) != 0.) {
            if (pv != c0)
                for (int m = 0; m < nr; ++m) { const double tmp = A(c0 + nr * m); A(c0 + nr * m) = A(pv + nr * m); A(pv + nr * m) = tmp; }
            const double r = 1. / A(c0 + nr * c0);
            for (int i = c0 + 1; i < nr; ++i) A(i + nr * c0) *= r;
        }
        for (int m = c0 + 1; m < nr; ++m) {
            const double u = A(c0 + nr * m);
            for (int i = c0 + 1; i < nr; ++i) A(i + nr * m) -= A(i + nr * c0) * u;
        }
    }
    if (dbg.LU) {
        for (int e = 0; e < nr * nr; ++e) dbg.LU[j * dbg.mat_stride + e] = A(e);
        for (int i = 0; i < nr; ++i) {
            dbg.row_scale[j * dbg.vec_stride + i] = V(RS, i); dbg.col_scale[j * dbg.vec_stride + i] = V(CS, i);
            dbg.ipiv[j * dbg.vec_stride + i] = (int)V(DR, i);
        }
    }
    // dgetrs('N') (lapackdrivers.pyx:1657-1665) on the vector BB
    auto lu_solve = [&]() {
        for (int i = 0; i < nr; ++i) {
            const int pv = (int)V(DR, i) - 1;
            if (pv != i) { const double tmp = V(BB, i); V(BB, i) = V(BB, pv); V(BB, pv) = tmp; }
        }
        for (int c0 = 0; c0 < nr; ++c0) {
            const double bj = V(BB, c0);
            for (int i = c0 + 1; i < nr; ++i) V(BB, i) -= A(i + nr * c0) * bj;
        }
        for (int c0 = nr - 1; c0 >= 0; --c0) {
            const double bj = V(BB, c0) / A(c0 + nr * c0);
            V(BB, c0) = bj;
            for (int i = 0; i < c0; ++i) V(BB, i) -= A(i + nr * c0) * bj;
        }
    };

    // ---- solve (impl.pyx:731-846).  `rhs(k)` is fk[k] (first solve) or the residual (refinement); `fin(om)` the known values.
    // Leaves the reduced solution b[jj] * col_scale[jj] in BB.
    auto solve = [&](auto&& rhs, auto&& fin) {
        for (int jj = 0; jj < nr; ++jj) V(BB, jj) = 0.;
        for (int k = 0; k < nk; ++k) {
            double d[DIM], c[NO];
            rows.offset(k, xi, d);
            const double d2 = make_c<DIM, ORDER>(d, c);
            const double wf = make_weight(d2, max_d2, uniform) * rhs(k, d);
            // static walk over the DOFs keeps c[] in registers; jj counts the unknowns passed so far
            int jj = 0;
#pragma unroll
            for (int a = 0; a < NO; ++a) {
                if (!((knowns >> a) & 1ll) && jj < nr) { V(BB, jj) += wf * c[a]; ++jj; }
            }
        }
        for (int jj = 0; jj < nr; ++jj) V(BB, jj) = V(RS, jj) * V(BB, jj);
        // knowns move to the right-hand side, term by term into b[jj] (impl.pyx:792-818)
#pragma unroll 1
        for (int om = 0; om < NO; ++om) {
            if (!((knowns >> om) & 1ll)) continue;
            const double fom = fin(om);
            for (int k = 0; k < nk; ++k) {
                double d[DIM], c[NO];
                rows.offset(k, xi, d);
                const double d2 = make_c<DIM, ORDER>(d, c);
                const double w = make_weight(d2, max_d2, uniform);
                double com = 0.;
#pragma unroll
                for (int a = 0; a < NO; ++a) if (a == om) com = c[a];
                const double fwc = fom * w * com;
                int jj = 0;
#pragma unroll
                for (int a = 0; a < NO; ++a) {
                    if (!((knowns >> a) & 1ll) && jj < nr) { V(BB, jj) -= fwc * c[a] * V(RS, jj); ++jj; }
                }
            }
        }
        lu_solve();
        for (int jj = 0; jj < nr; ++jj) V(BB, jj) = V(BB, jj) * V(CS, jj);
    };

    solve([&](int k, const double (&)[DIM]) { return rows.value(k); }, [&](int om) { return fio[om]; });

    // ---- sensitivities (impl.pyx:776-778, 821-846): one dgetrs per neighbour
    if (p.do_sens && p.sens) {
        // the first solve's result must survive: park it in WFI
        for (int jj = 0; jj < nr; ++jj) V(WFI, jj) = V(BB, jj);
        double* const sr = p.sens + j * p.ss_j;
        for (int k = 0; k < nk; ++k) {
            double d[DIM], c[NO];
            rows.offset(k, xi, d);
            const double d2 = make_c<DIM, ORDER>(d, c);
            const double w = make_weight(d2, max_d2, uniform);
            int jj = 0;
#pragma unroll
            for (int a = 0; a < NO; ++a) {
                if (!((knowns >> a) & 1ll) && jj < nr) { V(BB, jj) = V(RS, jj) * w * c[a]; ++jj; }
            }
            lu_solve();
            for (int q = 0; q < nr; ++q) sr[k * p.ss_k + (int)V(R2O, q)] = V(BB, q) * V(CS, q);
            for (int om = 0; om < NO; ++om)
                if ((knowns >> om) & 1ll) sr[k * p.ss_k + om] = __longlong_as_double(0x7ff8000000000000LL);
        }
        for (int jj = 0; jj < nr; ++jj) V(BB, jj) = V(WFI, jj);
    }

    if (!p.iterative) {
        for (int q = 0; q < nr; ++q) fio[(int)V(R2O, q)] = V(BB, q);
        return;
    }

    // ---- solve_iterative (impl.pyx:986-1083): fi = the case's copy of the user's row with the unknowns just solved
    for (int a = 0; a < NO; ++a) V(FI, a) = fio[a];
    for (int q = 0; q < nr; ++q) V(FI, (int)V(R2O, q)) = V(BB, q);
    double prev_norm = -1.;
    bool broke = false;
    int i = 0;
    for (i = 0; i < p.max_iter; ++i) {
        // the residual's maximum norm first (impl.pyx:1037-1057): exact equality with the previous one stops the loop
        double norm = 0.;
        for (int k = 0; k < nk; ++k) {
            double d[DIM];
            rows.offset(k, xi, d);
            const double res = rows.value(k) - taylor<DIM, ORDER>(d, [&](int a) { return V(FI, a); });
            const double ar = fabs(res);
            if (k == 0) norm = ar; else if (ar > norm) norm = ar;
        }
        if (norm == prev_norm) { broke = true; break; }
        prev_norm = norm;
        // the correction: the same solve on the residual, knowns of the correction are zero (impl.pyx:1015-1017)
        solve([&](int k, const double (&d)[DIM]) { return rows.value(k) - taylor<DIM, ORDER>(d, [&](int a) { return V(FI, a); }); },
              [&](int) { return 0.; });
        for (int q = 0; q < nr; ++q) { const int a = (int)V(R2O, q); V(FI, a) = V(FI, a) + V(BB, q); }
    }
    const int iters = broke ? i : (p.max_iter > 0 ? p.max_iter : 1);      // for/else, impl.pyx:1080-1081
    for (int q = 0; q < nr; ++q) { const int a = (int)V(R2O, q); fio[a] = V(FI, a); }
    if (p.iters_out) atomicMax(p.iters_out, iters);
}

template <int DIM, int ORDER>
__global__ __launch_bounds__(64) void fit_strict_kernel(const KParams p, const StrictDebug dbg, const int skip_plain_groups) {
    extern __shared__ double smem[];
    fit_strict_block<DIM, ORDER>(p, dbg, skip_plain_groups, blockIdx.x, smem);
}

// The same operations with EVERYTHING in registers, for the common case of a workgroup whose 64 cases have no knowns at all (the
// reduced system is the full one: every index is a compile-time constant) and a basic fit: no LDS, so the occupancy is set by
// the registers (two or more waves per SIMD) instead of by 48 KB of LDS per 64 cases (three waves per CU) — BASELINE configs[1]
// in strict mode 2.5 -> ~0.5 ms per 1M fits.  The pivot row of a lane is data-dependent, registers cannot be indexed by it: the
// row exchange is written as selects over the candidate rows (N^3 / 3 of them: nothing against the ~8 Ruiz sweeps of N^2 IEEE
// divides).  A workgroup with any known DOF, sensitivities, refinement or the debug capture runs the LDS kernel above; both
// kernels are launched over the same 64-case groups and each group is taken by exactly one of them (fit_strict_group_is_plain).
__device__ __forceinline__ bool fit_strict_group_is_plain(const KParams& p, long long t, long long ncases, long long want) {
    // (block-uniform result; every thread of the 64-thread block must call it): every case of the group has knowns == want
    bool mine = false;
    if (t < ncases) {
        const long long j = p.case_index ? p.case_index[t] : t;
        mine = p.knowns[j * p.sknowns] != want;
    }
    return __syncthreads_or(mine ? 1 : 0) == 0;
}

#ifndef WLSQM_STRICT_REG_MINW6
#define WLSQM_STRICT_REG_MINW6 1      // waves per SIMD the register kernel of the systems up to 6 unknowns is compiled for: registers as needed (C2: 0.74 ms; capped for three / four waves it spills: 0.95 / 1.29; profiles/r03j_ab_strict_minw.txt)
#endif
#ifndef WLSQM_STRICT_REG_MINW10
#define WLSQM_STRICT_REG_MINW10 1     // ... of the 10-unknown systems (C5: 2.00 ms; capped for two waves: 2.79)
#endif
__host__ __device__ constexpr int reg_minw(int NO) { return NO <= 6 ? WLSQM_STRICT_REG_MINW6 : WLSQM_STRICT_REG_MINW10; }

// KN1: the groups whose 64 cases all have exactly the function value known (knowns = b?_F = 1, the reference's default mask): the
// reduced system is DOFs 1 .. NO - 1, again with compile-time indices; the known value moves to the right-hand side term by term
// in a third pass over the neighbours (it needs the row scales: impl.pyx:815-818 multiplies every term by row_scale[j]).
template <int DIM, int ORDER, bool KN1>
__device__ __forceinline__ void fit_strict_reg_block(const KParams& p, const long long vblock) {
    using namespace strict;
    constexpr int NO = ndofs(DIM, ORDER);
    constexpr int N = NO - (KN1 ? 1 : 0), O0 = KN1 ? 1 : 0;      // reduced size; reduced index i is DOF i + O0
    const long long ncases = live_cases(p);
    const long long t = vblock * 64 + threadIdx.x;
    if (!fit_strict_group_is_plain(p, t, ncases, KN1 ? 1 : 0)) return;     // another kind of group: another kernel has it
    if (t >= ncases) return;
    const long long j = p.case_index ? p.case_index[t] : t;
    const int nk = min(p.nk[j * p.snk], (int)p.max_nk);
    const bool uniform = (p.wm[j * p.swm] == WLSQM_WEIGHT_UNIFORM);
    double xi[DIM];
    Rows<DIM> rows;
    if (p.hoods) {
        const long long pj = own_point(p, j);
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.S[pj * DIM + m];
        rows = Rows<DIM>{nullptr, 0, nullptr, 0, p.hoods + j * p.shoods_j, p.S, p.F};
    } else {
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.xi[j * p.sxi_j + m];
        rows = Rows<DIM>{p.xk + j * p.sxk_j, p.sxk_k, p.fk + j * p.sfk_j, p.sfk_k, nullptr, nullptr, nullptr};
    }
    // Round 4: the quotients and roots of the weights and of the equilibration run the compiler's IEEE sequences WITHOUT their range
    // scaling and special-case fix-up (acc::FastOps, wlsqm_strict.hpp: the same operations, the same bits; seeded reciprocals in the
    // equilibration) when every operand of the wave's cases is in their safe range — checked here —, and the full sequences otherwise.
    double max_d2 = 0., min_d2 = acc::RANGE_HI;
    for (int k = 0; k < nk; ++k) {
        double d[DIM], c[NO];
        rows.offset(k, xi, d);
        const double d2 = make_c<DIM, ORDER>(d, c);
        if (d2 > max_d2) max_d2 = d2;
        if (!(d2 >= min_d2)) min_d2 = d2;                        // (a NaN distance lands here and fails the range test)
    }
    const bool fast_w = __all(uniform || (nk > 0 && min_d2 >= acc::RANGE_LO && max_d2 <= acc::RANGE_HI));
    // make_A (impl.pyx:566-602) and the right-hand side sums of solve (impl.pyx:768-787): each its own sum over k ascending
    double A[N][N], b[N];                                        // A[row][col]
#pragma unroll
    for (int i = 0; i < N; ++i) { b[i] = 0.;
#pragma unroll
        for (int m = 0; m < N; ++m) A[i][m] = 0.; }
    auto weight_of = [&](auto ops_tag, double d2, double rmax) {
        using OPS = decltype(ops_tag);
        if (uniform) return 1.;
        const double tmp = 1. - OPS::sqrt(OPS::div_r(d2, max_d2, rmax));
        return weights_alpha + weights_beta * tmp * tmp;
    };
    auto assemble = [&](auto ops_tag) {
        const double rmax = decltype(ops_tag)::rcp_of(max_d2);
        for (int k = 0; k < nk; ++k) {
            double d[DIM], c[NO];
            rows.offset(k, xi, d);
            const double d2 = make_c<DIM, ORDER>(d, c);
            const double w = weight_of(ops_tag, d2, rmax);
            const double wf = w * rows.value(k);
#pragma unroll
            for (int om = 0; om < N; ++om) {
                const double wc = w * c[om + O0];
#pragma unroll
                for (int oj = 0; oj < N; ++oj) A[oj][om] += wc * c[oj + O0];
            }
#pragma unroll
            for (int oj = 0; oj < N; ++oj) b[oj] += wf * c[oj + O0];
        }
    };
    if (fast_w) assemble(acc::FastOps{}); else assemble(acc::IeeeOps{});
    // rescale_ruiz2001_c (lapackdrivers.pyx:553-623)
    double rs[N], cs[N];
    auto ruiz = [&](auto ops_tag) -> bool {                       // returns: every running scale factor stayed in the safe range
        using OPS = decltype(ops_tag);
        double DRp[N], DCp[N], DR[N], DC[N];
        bool in_range = true;
#pragma unroll
        for (int i = 0; i < N; ++i) { rs[i] = 1.; cs[i] = 1.; DRp[i] = 1.; DCp[i] = 1.; }
        for (int it = 0; it < 100; ++it) {
#pragma unroll
            for (int i = 0; i < N; ++i) { DR[i] = 0.; DC[i] = 0.; }
#pragma unroll
            for (int m = 0; m < N; ++m) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    // (fast path: rs[i] cs[m] is within 2^-48 of the reciprocal of DRp[i] DCp[m] — every sweep multiplies the one
                    // and divides the other by the same root —, the seed of the quotient's reciprocal; first sweep: 1 x 1)
                    const double q = fabs(OPS::div_seeded(A[i][m], DRp[i] * DCp[m], rs[i] * cs[m]));
                    DC[m] = OPS::maxnum(DC[m], q);
                    DR[i] = OPS::maxnum(DR[i], q);
                }
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double hr, hc;
                DR[i] = OPS::sqrt_h(DR[i], hr); DC[i] = OPS::sqrt_h(DC[i], hc);
                DRp[i] *= DR[i]; rs[i] = OPS::div_by_root(rs[i], DR[i], hr);
                DCp[i] *= DC[i]; cs[i] = OPS::div_by_root(cs[i], DC[i], hc);
                in_range = in_range && DRp[i] >= acc::SCALE_LO && DRp[i] <= acc::SCALE_HI && DCp[i] >= acc::SCALE_LO && DCp[i] <= acc::SCALE_HI;
            }
            double accm = fabs(1. - DR[0] * DR[0]);
#pragma unroll
            for (int i = 1; i < N; ++i) { const double tmp = fabs(1. - DR[i] * DR[i]); if (tmp > accm) accm = tmp; }
            if (accm < ruiz_epsilon) {
                accm = fabs(1. - DC[0] * DC[0]);
#pragma unroll
                for (int i = 1; i < N; ++i) { const double tmp = fabs(1. - DC[i] * DC[i]); if (tmp > accm) accm = tmp; }
                if (accm < ruiz_epsilon) break;
            }
        }
        return in_range;
    };
    {
        bool r_ok = true, fast_done = false;                      // every nonzero entry in the safe range and no zero row or column
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double rowmax = 0., colmax = 0.;
#pragma unroll
            for (int m = 0; m < N; ++m) {
                const double a = fabs(A[i][m]), t2 = fabs(A[m][i]);
                r_ok = r_ok && (a == 0. || (a >= acc::RANGE_LO && a <= acc::RANGE_HI));
                rowmax = a > rowmax ? a : rowmax; colmax = t2 > colmax ? t2 : colmax;
            }
            r_ok = r_ok && rowmax >= acc::RANGE_LO && colmax >= acc::RANGE_LO;
        }
        if (__all(r_ok)) { r_ok = ruiz(acc::FastOps{}); fast_done = true; }
        if (!fast_done || !__all(r_ok)) (void)ruiz(acc::IeeeOps{});
    }
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
        for (int i = 0; i < N; ++i) A[i][m] *= (rs[i] * cs[m]);           // apply_scaling_c (lapackdrivers.pyx:293-299)
    // dgetrf (unblocked dgetf2 semantics), the row exchange as selects
    int ipiv[N];
#pragma unroll
    for (int c0 = 0; c0 < N; ++c0) {
        int pv = c0; double best = fabs(A[c0][c0]), pval = A[c0][c0];
#pragma unroll
        for (int i = c0 + 1; i < N; ++i) { const double v = fabs(A[i][c0]); if (v > best) { best = v; pv = i; pval = A[i][c0]; } }
        ipiv[c0] = pv;
        if (pval != 0.) {
#pragma unroll
            for (int i = c0 + 1; i < N; ++i) {
                const bool sw = (pv == i);
#pragma unroll
                for (int m = 0; m < N; ++m) { const double u = A[c0][m], v = A[i][m]; A[c0][m] = sw ? v : u; A[i][m] = sw ? u : v; }
            }
            const double r = 1. / A[c0][c0];
#pragma unroll
            for (int i = c0 + 1; i < N; ++i) A[i][c0] *= r;
        }
#pragma unroll
        for (int m = c0 + 1; m < N; ++m) {
            const double u = A[c0][m];
#pragma unroll
            for (int i = c0 + 1; i < N; ++i) A[i][m] -= A[i][c0] * u;
        }
    }
    // solve (impl.pyx:731-846) without knowns: b = row_scale * sums, dgetrs('N'), un-scale
    double* const fio = p.fi + j * p.sfi_j;
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] = rs[i] * b[i];
    if constexpr (KN1) {
        // the known function value moves to the right-hand side, term by term into b[j] (impl.pyx:792-818)
        const double f0 = fio[0];
        const double rmax3 = fast_w ? acc::rcp_refined(max_d2) : 0.;
        for (int k = 0; k < nk; ++k) {
            double d[DIM], c[NO];
            rows.offset(k, xi, d);
            const double d2 = make_c<DIM, ORDER>(d, c);
            const double w = fast_w ? weight_of(acc::FastOps{}, d2, rmax3) : weight_of(acc::IeeeOps{}, d2, 0.);
            const double fwc = f0 * w * c[0];
#pragma unroll
            for (int i = 0; i < N; ++i) b[i] -= fwc * c[i + O0] * rs[i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int q = i + 1; q < N; ++q) { const bool sw = (ipiv[i] == q); const double u = b[i], v = b[q]; b[i] = sw ? v : u; b[q] = sw ? u : v; }
    }
#pragma unroll
    for (int c0 = 0; c0 < N; ++c0)
#pragma unroll
        for (int i = c0 + 1; i < N; ++i) b[i] -= A[i][c0] * b[c0];
#pragma unroll
    for (int c0 = N - 1; c0 >= 0; --c0) {
        b[c0] /= A[c0][c0];
#pragma unroll
        for (int i = 0; i < c0; ++i) b[i] -= A[i][c0] * b[c0];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) fio[i + O0] = b[i] * cs[i];
}

template <int DIM, int ORDER, bool KN1>
__global__ __launch_bounds__(64, reg_minw(ndofs(DIM, ORDER))) void fit_strict_reg_kernel(const KParams p) {
    fit_strict_reg_block<DIM, ORDER, KN1>(p, blockIdx.x);
}

constexpr int STRICT_REG_MAX_NO = 10;      // register kernel: systems up to this size (3D order 2 / 2D order 3: one wave per SIMD)

// ROW PER LANE: the same operations for the systems too large for the register kernel and for every call with knowns,
// sensitivities or refinement.  A case is handled by LPC = 16 / 32 / 64 lanes; lane i of the group owns ROW i of the reduced
// matrix (A[m] = entry (i, m), compile-time m) and element i of every per-case vector, so the matrix of a 14 x 14 system is 14
// doubles per lane instead of an LDS image of 196 per case (which allowed 16 cases per workgroup: 74 ms per 1M fits of BASELINE
// configs[2]).  No sum is split: entry (i, m) is still one sum over k ascending in one lane, and the only cross-lane steps are
// order-free (maxima, with the reference's NaN rules spelled out below) or plain data movement: the column maxima of the
// equilibration, the pivot search (first maximum wins), the exchange of two rows, and the broadcast of a pivot row / a solution
// element.  The neighbours of a group (offsets, values, weights — each computed once, neighbour k by lane k mod LPC) sit in LDS
// and are read back as broadcasts.  Loops whose trip count differs between the groups of a wave (neighbours, equilibration
// sweeps, refinement passes) run to the wave's maximum with the finished groups frozen, because every lane takes part in the
// shuffles.
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {       // unrolled in the front end: indices are constants before any pass runs
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
template <int LPC>
__device__ __forceinline__ double grp_get(double v, int src) { return __shfl(v, src, LPC); }
// Lane permutations inside a row of 16 lanes as DPP moves (vector ALU, no LDS pipe)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// maximum over the group for values that are never NaN (callers map NaN away first)
template <int LPC>
__device__ __forceinline__ double grp_max(double v) {
    if constexpr (LPC >= 2) v = fmax(v, dpp_move<0xB1>(v));        // quad_perm [1,0,3,2]
    if constexpr (LPC >= 4) v = fmax(v, dpp_move<0x4E>(v));        // quad_perm [2,3,0,1]
    if constexpr (LPC >= 8) v = fmax(v, dpp_move<0x141>(v));       // row_half_mirror: lane i <-> 7 - i of each 8
    if constexpr (LPC >= 16) v = fmax(v, dpp_move<0x140>(v));      // row_mirror: lane i <-> 15 - i of each 16
#pragma unroll
    for (int s = 16; s < LPC; s <<= 1) v = fmax(v, __shfl_xor(v, s, LPC));
    return v;
}
// Waves per SIMD the row kernel is compiled for.  It is bound by the issue of dependent vector instructions (IEEE divides, selects), so a
// third / fourth resident wave pays for a few spilled registers: 1M cases, 2D order 4 (F known): 9.5 ms as allocated freely (190
// VGPRs), 7.97 at three waves (168 + 33 spilled), 8.03 at four; with sensitivities 28.6 / 23.0 / 25.9; 3D order 2 with sensitivities:
// 9.37 / 9.37 / 8.63 (profiles/r03j_ab_strict_minw.txt).  The 20- and 35-unknown systems keep their registers (242 / 389).
__host__ __device__ constexpr int rows_minw(int NO) {
    return WLSQM_STRICT_ROWS_MINW ? WLSQM_STRICT_ROWS_MINW : (NO > 16 ? 1 : NO > 10 ? 3 : 4);
}

// (vblock: the workgroup's number in the batch)
template <int DIM, int ORDER, int LPC>
__device__ __forceinline__ void fit_strict_rows_block(const KParams& p, const int KP, const long long vblock, double* smem) {
    using namespace strict;
    constexpr int NO = ndofs(DIM, ORDER);
    constexpr int G = 64 / LPC;                       // cases per wave
    static_assert(NO <= LPC, "one lane per row");
    const int lane = threadIdx.x, g = lane / LPC, i = lane % LPC;
    const long long ncases = live_cases(p);
    const long long t = vblock * G + g;
    const bool valid = t < ncases;
    const long long j = valid ? (p.case_index ? p.case_index[t] : t) : 0;
    // LDS of the group: w[KP], f[KP], res[KP], d[DIM][KP]; the pitch is odd in 8-byte words so that the G broadcast reads of one
    // instruction fall into different banks
    const int pitch = ((3 + DIM) * KP) | 1;
    double* const sw = smem + (size_t)g * pitch;
    double* const sf = sw + KP;
    double* const sres = sf + KP;
    double* const sd = sres + KP;

    const int nk = valid ? min(p.nk[j * p.snk], (int)p.max_nk) : 0;
    const bool uniform = valid && (p.wm[j * p.swm] == WLSQM_WEIGHT_UNIFORM);
    const long long knowns = valid ? p.knowns[j * p.sknowns] : 0;
    const int nr = NO - __popcll((unsigned long long)knowns);            // infra.pyx:119-121: bits >= no are not masked
    const bool row = i < nr;                                             // this lane owns a row of the reduced system
    // infra.pyx:145-200 (remap): reduced index i -> DOF, unknowns in ascending DOF order
    int dof = 0;
    {
        int cnt = 0;
#pragma unroll
        for (int a = 0; a < NO; ++a) {
            const bool unk = !((knowns >> a) & 1ll);
            if (unk && cnt == i) dof = a;
            cnt += unk ? 1 : 0;
        }
    }
    int nkmax = nk;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) nkmax = max(nkmax, __shfl_xor(nkmax, s, 64));

    double xi[DIM];
    Rows<DIM> rows;
    if (p.hoods) {
        const long long pj = own_point(p, j);
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.S[pj * DIM + m];
        rows = Rows<DIM>{nullptr, 0, nullptr, 0, p.hoods + j * p.shoods_j, p.S, p.F};
    } else {
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.xi[j * p.sxi_j + m];
        rows = Rows<DIM>{p.xk + j * p.sxk_j, p.sxk_k, p.fk + j * p.sfk_j, p.sfk_k, nullptr, nullptr, nullptr};
    }
    double* const fio = p.fi + j * p.sfi_j;

    // ---- make_c pass 1 (neighbour k by lane k mod LPC): offsets and values into LDS, largest squared distance.  The reference's
    // `if d2 > max_d2` skips NaNs and so does every step of the maximum here.
    double max_d2 = 0.;
    for (int k = i; k < nk; k += LPC) {
        double d[DIM], c[NO];
        rows.offset(k, xi, d);
        const double d2 = make_c<DIM, ORDER>(d, c);
        if (d2 > max_d2) max_d2 = d2;
#pragma unroll
        for (int m = 0; m < DIM; ++m) sd[m * KP + k] = d[m];
        sf[k] = rows.value(k);
    }
    max_d2 = grp_max<LPC>(max_d2);
    for (int k = i; k < nk; k += LPC) {
        double d[DIM], c[NO];
#pragma unroll
        for (int m = 0; m < DIM; ++m) d[m] = sd[m * KP + k];
        const double d2 = make_c<DIM, ORDER>(d, c);
        sw[k] = make_weight(d2, max_d2, uniform);
    }
    __syncthreads();

    // ---- make_A (impl.pyx:566-602), row `dof` of the FULL index set: entry (dof, om) = sum_k (w c_om) c_dof, k ascending, and the
    // right-hand side sum of solve (impl.pyx:768-787); the reduced row is what is left after the known columns are deleted
    double A[NO], b = 0.;
#pragma unroll
    for (int m = 0; m < NO; ++m) A[m] = 0.;
    for (int k = 0; k < nkmax; ++k) {
        if (k < nk) {
            double d[DIM], c[NO];
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = sd[m * KP + k];
            make_c<DIM, ORDER>(d, c);
            const double w = sw[k];
            const double cown = pick<NO>(c, dof);
#pragma unroll
            for (int om = 0; om < NO; ++om) { const double wc = w * c[om]; A[om] += wc * cown; }
            const double wf = w * sf[k];
            b += wf * cown;
        }
    }
#pragma unroll
    for (int a = NO - 1; a >= 0; --a) {          // delete column a where DOF a is known (from the top: the lower ones stay put)
        const bool kn = (knowns >> a) & 1ll;
#pragma unroll
        for (int m = a; m < NO - 1; ++m) A[m] = kn ? A[m + 1] : A[m];
    }
#pragma unroll
    for (int m = 0; m < NO; ++m) A[m] = (row && m < nr) ? A[m] : 0.;     // beyond the reduced system: zeros (never win a maximum)

    // ---- rescale_ruiz2001_c (lapackdrivers.pyx:553-623) with init_scaling_c (:285-290).  Lane i keeps the row quantities of row i
    // and the column quantities of column i.  q is the same double in the reference's row pass and column pass.
    double rs = 1., cs = 1., DRp = 1., DCp = 1.;
    {
        bool live = valid && nr >= 1;
        for (int it = 0; it < 100; ++it) {
            double rowmax = 0., colmine = 0.;
#pragma unroll
            for (int m = 0; m < NO; ++m) {
                const double dcp = grp_get<LPC>(DCp, m);
                const double q = fabs(A[m] / (DRp * dcp));
                if (q > rowmax) rowmax = q;
                const double cm = grp_max<LPC>(q > 0. ? q : 0.);         // NaN -> 0: skipped like the reference's `q > cmax`
                if (i == m) colmine = cm;
            }
            const double DR = row ? sqrt(rowmax) : 1., DC = row ? sqrt(colmine) : 1.;
            if (live) { DRp *= DR; rs /= DR; DCp *= DC; cs /= DC; }
            // stop test: max_i |1 - DR_i^2| < eps and the same for DC; the scan starts from element 0 and `tmp > acc` never replaces
            // a NaN there (then acc < eps is false): element 0's NaN counts as +inf, any other NaN is skipped
            double tr = fabs(1. - DR * DR), tc = fabs(1. - DC * DC);
            const double inf = __longlong_as_double(0x7ff0000000000000LL);
            if (tr != tr) tr = (i == 0) ? inf : 0.;
            if (tc != tc) tc = (i == 0) ? inf : 0.;
            const double acc = grp_max<LPC>(row ? (tr > tc ? tr : tc) : 0.);
            if (acc < ruiz_epsilon) live = false;
            if (!__any(live)) break;
        }
    }
#pragma unroll
    for (int m = 0; m < NO; ++m) { const double csm = grp_get<LPC>(cs, m); A[m] *= (rs * csm); }   // apply_scaling_c (:293-299)

    // ---- dgetrf (lapackdrivers.pyx:1628-1635; dgetf2: first maximal |a_ik|, column scaled by the reciprocal pivot)
    int ipiv[NO];
    static_for<0, NO>([&](auto C0) {
        constexpr int c0 = decltype(C0)::value;
        const bool step = c0 < nr;
        // candidates: rows c0 .. nr-1.  The scan keeps the first value unless a later one is GREATER: a NaN in row c0 stays (as
        // +inf here, the smaller index winning ties), a NaN elsewhere never wins (-1: below every |a|)
        double v = fabs(A[c0]);
        if (i < c0 || !row) v = -1.;
        else if (v != v) v = (i == c0) ? __longlong_as_double(0x7ff0000000000000LL) : -1.;
        int pv = i;
#pragma unroll
        for (int s = 1; s < LPC; s <<= 1) {
            const double ov = __shfl_xor(v, s, LPC);
            const int oi = __shfl_xor(pv, s, LPC);
            if (ov > v || (ov == v && oi < pv)) { v = ov; pv = oi; }
        }
        if (!step) pv = c0;
        ipiv[c0] = pv;
        const double pval = grp_get<LPC>(A[c0], pv);
        const bool nz = step && pval != 0.;
        const bool sw2 = nz && pv != c0;
        if (__any(sw2)) {
            const int src = (i == c0) ? pv : (i == pv) ? c0 : i;
#pragma unroll
            for (int m = 0; m < NO; ++m) { const double o = grp_get<LPC>(A[m], src); A[m] = sw2 ? o : A[m]; }
        }
        const double r = 1. / pval;
        const bool below = step && row && i > c0;
        if (nz && below) A[c0] *= r;
#pragma unroll
        for (int m = c0 + 1; m < NO; ++m) {
            const double u = grp_get<LPC>(A[m], c0);
            if (below) A[m] -= A[c0] * u;
        }
    });
    // dgetrs('N') (lapackdrivers.pyx:1657-1665) on the group's vector (element i in lane i)
    auto lu_solve = [&](double x) __attribute__((always_inline)) -> double {
        static_for<0, NO>([&](auto C0) {
            constexpr int c0 = decltype(C0)::value;
            const int pv = ipiv[c0];
            const bool sw2 = c0 < nr && pv != c0;
            if (__any(sw2)) {
                const int src = (i == c0) ? pv : (i == pv) ? c0 : i;
                const double o = grp_get<LPC>(x, src);
                x = sw2 ? o : x;
            }
        });
#pragma unroll
        for (int c0 = 0; c0 < NO; ++c0) {
            const double bj = grp_get<LPC>(x, c0);
            if (c0 < nr && row && i > c0) x -= A[c0] * bj;
        }
#pragma unroll
        for (int c0 = NO - 1; c0 >= 0; --c0) {
            if (i == c0) x = x / A[c0];
            const double bj = grp_get<LPC>(x, c0);
            if (c0 < nr && i < c0) x -= A[c0] * bj;
        }
        return x;
    };

    // ---- solve (impl.pyx:731-846): `sum` is this row's sum_k (w rhs_k) c_k, `fin(om)` the known values; returns b * col_scale
    auto finish = [&](double sum, auto&& fin) __attribute__((always_inline)) -> double {
        double x = rs * sum;
        // knowns move to the right-hand side, term by term (impl.pyx:792-818)
        static_for<0, NO>([&](auto OM) {
            constexpr int om = decltype(OM)::value;
            const bool kn = (knowns >> om) & 1ll;
            if (!__any(kn)) return;
            const double fom = kn ? fin(om) : 0.;
            for (int k = 0; k < nkmax; ++k) {
                if (kn && k < nk) {
                    double d[DIM], c[NO];
#pragma unroll
                    for (int m = 0; m < DIM; ++m) d[m] = sd[m * KP + k];
                    make_c<DIM, ORDER>(d, c);
                    const double fwc = fom * sw[k] * c[om];
                    x -= fwc * pick<NO>(c, dof) * rs;
                }
            }
        });
        x = lu_solve(x);
        return x * cs;
    };
    double x = finish(b, [&](int om) { return fio[om]; });

    // ---- sensitivities (impl.pyx:776-778, 821-846): one dgetrs per neighbour
    if (p.do_sens && p.sens) {
        double* const sr = p.sens + j * p.ss_j;
        for (int k = 0; k < nkmax; ++k) {
            double d[DIM], c[NO];
            const int kk = k < nk ? k : 0;
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = sd[m * KP + kk];
            make_c<DIM, ORDER>(d, c);
            double s = rs * sw[kk] * pick<NO>(c, dof);
            s = lu_solve(s);
            if (valid && nr >= 1 && k < nk) {
                if (row) sr[k * p.ss_k + dof] = s * cs;
                if (i < NO && ((knowns >> i) & 1ll)) sr[k * p.ss_k + i] = __longlong_as_double(0x7ff8000000000000LL);
            }
        }
    }

    if (!p.iterative) {
        if (valid && row) fio[dof] = x;
        return;
    }

    // ---- solve_iterative (impl.pyx:986-1083).  Every lane keeps the case's whole fi (the model needs all of it): the user's
    // row with the unknowns just solved.  Reduced index of DOF a = number of unknown DOFs below it.
    double FI[NO];
    auto spread = [&](double xr, bool add, bool apply) __attribute__((always_inline)) {       // every lane takes part in the shuffles; `apply` guards the update
        int cnt = 0;
#pragma unroll
        for (int a = 0; a < NO; ++a) {
            const bool unk = !((knowns >> a) & 1ll);
            const double o = grp_get<LPC>(xr, cnt < LPC ? cnt : 0);
            if (apply && unk && cnt < nr) FI[a] = add ? FI[a] + o : o;
            cnt += unk ? 1 : 0;
        }
    };
#pragma unroll
    for (int a = 0; a < NO; ++a) FI[a] = valid ? fio[a] : 0.;
    spread(x, false, true);
    double prev_norm = -1.;
    int iters = 0;
    bool live = valid && nr >= 1, broke = false;
    if (nr < 1) {
        // impl.pyx:1026-1081 with every solve a no-op: the second pass finds the first pass's norm and stops (count 1) — unless
        // the norm is NaN, which equals nothing: then all max_iter passes run.  It is NaN iff res[0] is.
        iters = 1;
        if (nk > 0 && p.max_iter > 1) {
            double d0[DIM];
#pragma unroll
            for (int m = 0; m < DIM; ++m) d0[m] = sd[m * KP + 0];
            const double res0 = sf[0] - taylor<DIM, ORDER>(d0, [&](int a) { return FI[a]; });
            if (res0 != res0) iters = p.max_iter;
        }
    }
    for (int it = 0; it < p.max_iter; ++it) {
        if (!__any(live)) break;
        // residuals (neighbour k by lane k mod LPC) and their maximum norm (impl.pyx:1037-1057): the scan starts from res[0] and
        // `ar > norm` neither replaces nor picks a NaN
        double norm = 0.;
        bool nan0 = false;
        for (int k = i; k < nk; k += LPC) {
            double d[DIM];
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = sd[m * KP + k];
            const double res = sf[k] - taylor<DIM, ORDER>(d, [&](int a) { return FI[a]; });
            sres[k] = res;
            const double ar = fabs(res);
            if (k == 0 && ar != ar) nan0 = true;
            if (ar > norm) norm = ar;
        }
        norm = grp_max<LPC>(norm);
        if (__shfl(nan0 ? 1 : 0, 0, LPC)) norm = __longlong_as_double(0x7ff8000000000000LL);
        __syncthreads();
        if (live && norm == prev_norm) { live = false; broke = true; iters = it; }
        prev_norm = norm;
        // the correction: the same solve on the residual, knowns of the correction are zero (impl.pyx:1015-1017)
        double sum = 0.;
        for (int k = 0; k < nkmax; ++k) {
            if (k < nk) {
                double d[DIM], c[NO];
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = sd[m * KP + k];
                make_c<DIM, ORDER>(d, c);
                const double wf = sw[k] * sres[k];
                sum += wf * pick<NO>(c, dof);
            }
        }
        const double dx = finish(sum, [&](int) { return 0.; });
        spread(dx, true, live);
        __syncthreads();
    }
    if (nr >= 1) iters = broke ? iters : (p.max_iter > 0 ? p.max_iter : 1);      // for/else, impl.pyx:1080-1081
    if (valid && row) fio[dof] = pick<NO>(FI, dof);
    if (valid && i == 0 && p.iters_out) atomicMax(p.iters_out, iters);
}

template <int DIM, int ORDER, int LPC>
__global__ __launch_bounds__(64, rows_minw(ndofs(DIM, ORDER))) void fit_strict_rows_kernel(const KParams p, const int KP) {
    extern __shared__ double smem[];
    fit_strict_rows_block<DIM, ORDER, LPC>(p, KP, blockIdx.x, smem);
}

template <int DIM, int ORDER>
static int launch_strict(const KParams& p, const StrictDebug& dbg, hipStream_t stream) {
    constexpr int NO = ndofs(DIM, ORDER);
    constexpr int LPW = strict::lanes_for(NO);
    constexpr size_t lds = (size_t)strict::slots(NO) * LPW * sizeof(double);
    static_assert(lds <= 160 * 1024, "strict image exceeds the LDS of a CU");
    const long long blocks = (p.ncases + LPW - 1) / LPW;
    if (blocks <= 0) return WLSQM_OK;
    if (blocks > 0x7fffffffLL) { set_error("too many cases for one launch"); return WLSQM_EVALUE; }
    if (lds > 64 * 1024) {
        static std::atomic<bool> optin[16] = {};   // idempotent opt-in: a race only repeats it
        int dev = 0;
        WLSQM_HIP_CHECK(hipGetDevice(&dev));
        if (dev >= 0 && dev < 16 && !optin[dev]) {
            WLSQM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fit_strict_kernel<DIM, ORDER>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            optin[dev] = true;
        }
    }
    // row-per-lane kernel: the larger systems always, the small ones whenever the register kernel does not apply to the call
    // (sensitivities, refinement) — unless the neighbour rows do not fit the LDS or the intermediates are captured
    {
        constexpr int LPC = NO <= 2 ? 2 : NO <= 4 ? 4 : NO <= 8 ? 8 : NO <= 16 ? 16 : NO <= 32 ? 32 : 64;
        const char* e = getenv("WLSQM_HIP_STRICT_NO_ROWS");
        const int KP = p.max_nk > 0 ? (int)p.max_nk : 1;
        const size_t rl = (size_t)((((3 + DIM) * KP) | 1) * (64 / LPC)) * sizeof(double);
        const bool want = NO > STRICT_REG_MAX_NO || p.do_sens || p.iterative;
        if (want && !dbg.A && !dbg.w && !dbg.LU && rl <= 160 * 1024 && !(e && e[0] == '1')) {
            if (rl > 64 * 1024) {
                WLSQM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fit_strict_rows_kernel<DIM, ORDER, LPC>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl));
            }
            const long long wgs = (p.ncases + (64 / LPC) - 1) / (64 / LPC);
            if (wgs > 0x7fffffffLL) { set_error("too many cases for one launch"); return WLSQM_EVALUE; }
            hipLaunchKernelGGL((fit_strict_rows_kernel<DIM, ORDER, LPC>), dim3((unsigned)wgs), dim3(64), rl, stream, p, KP);
            WLSQM_HIP_CHECK(hipGetLastError());
            note_kernel("strict-rows");
            return WLSQM_OK;
        }
    }
    // basic fits of the small systems: the all-unknown 64-case groups run the register kernel, the others the LDS kernel
    bool split = false;
    if constexpr (NO <= STRICT_REG_MAX_NO) {
        const char* e = getenv("WLSQM_HIP_STRICT_NO_REG");
        split = !p.do_sens && !p.iterative && !dbg.A && !dbg.w && !(e && e[0] == '1');
        if (split) {
            const long long groups = (p.ncases + 63) / 64;
            hipLaunchKernelGGL((fit_strict_reg_kernel<DIM, ORDER, false>), dim3((unsigned)groups), dim3(64), 0, stream, p);
            WLSQM_HIP_CHECK(hipGetLastError());
            if constexpr (NO >= 2) {
                hipLaunchKernelGGL((fit_strict_reg_kernel<DIM, ORDER, true>), dim3((unsigned)groups), dim3(64), 0, stream, p);
                WLSQM_HIP_CHECK(hipGetLastError());
            }
        }
    }
    hipLaunchKernelGGL((fit_strict_kernel<DIM, ORDER>), dim3((unsigned)blocks), dim3(64), lds, stream, p, dbg, split ? 1 : 0);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("strict");
    return WLSQM_OK;
}

int launch_fit_accurate(int dimension, int order, const KParams& p, hipStream_t stream, bool* handled);      // fit_accurate.hip

int launch_fit_strict(int dimension, int order, const KParams& p, const StrictDebug* dbg_in, hipStream_t stream) {
    const StrictDebug dbg = dbg_in ? *dbg_in : StrictDebug{};
    // accurate mode: fit_accurate.hip fits EVERY case of a basic call on the 2D / 3D systems up to 10 unknowns, in one launch; every other
    // call of the mode (2D order 4, 3D orders 3-4, 1D, sensitivities, refinement, the debug capture) is the strict mode's
    if (accurate_mode() && !dbg_in) {
        bool taken = false;
        const int rc = launch_fit_accurate(dimension, order, p, stream, &taken);
        if (rc != WLSQM_OK) return rc;
        if (taken) { note_kernel("accurate"); return WLSQM_OK; }
    }
#define CASE(D, O) if (dimension == D && order == O) return launch_strict<D, O>(p, dbg, stream);
    CASE(1, 0) CASE(1, 1) CASE(1, 2) CASE(1, 3) CASE(1, 4)
    CASE(2, 0) CASE(2, 1) CASE(2, 2) CASE(2, 3) CASE(2, 4)
    CASE(3, 0) CASE(3, 1) CASE(3, 2) CASE(3, 3) CASE(3, 4)
#undef CASE
    set_error("fit_strict: unsupported (dimension, order)");
    return WLSQM_EVALUE;
}

}  // namespace wlsqm
