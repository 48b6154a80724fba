// fit_tilek.hip — the LDS-tiled fit (fit_tile.hip) for ANY neighbour-axis extent K.
//
// fit_tile.hip is specialised per (dim, order, K) for the benchmark shapes; a user batch with
// K = 20, 25, 30, 50 ... would otherwise fall to the generic row-per-lane kernel at ~1/6 of the speed.
// This variant keeps K at run time: row strides, chunk counts and the neighbour split are computed by
// the host, staging runs in rounds of R 16-byte chunks per lane (all R loads of a round in flight), and
// the neighbour loops are rolled.  Same arithmetic and the same parity as fit_tile / fit_lane.
// Mapping: 4 waves per 64-case tile, wave w owns neighbours [w*KPW, (w+1)*KPW).
// MOM: accumulate the distinct moments (wlsqm_moments.hpp) instead of the matrix entries; wave 0 expands them.
#include <cstdlib>

#include <cstdio>
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_moments.hpp"
#include "wlsqm_tile1.hpp"

#ifndef WLSQM_TILEK_SENS_NT
#define WLSQM_TILEK_SENS_NT 1      // non-temporal stores of the sensitivities' slabs (written once): 1M cases with do_sens, 3D order 2 / 40: 1.62-1.64 against 1.72-1.74 ms; 2D order 2 / 32: 0.61-0.65 against 0.63-0.66 (profiles/r03i_ab_tilek_nt.txt)
#endif

namespace wlsqm {

constexpr int KW = 64, KSP = 4, KNT = KW * KSP, KROUND = 6;
typedef double kd2_ __attribute__((ext_vector_type(2)));

struct TileKGeom {
    int K, KPW;            // neighbour slots per case; per wave
    int RS, FS;            // LDS row strides (doubles)
    int XCH, FCH;          // 16-byte chunks of the tile's xk / fk blocks
    int CPRX, CPRF;        // chunks per row
    int lds_main;          // doubles shared by the tile image and the reduction buffer
};

template <int DIM, int ORDER, bool MOM>
__global__ __launch_bounds__(KNT, 2) void fit_tilek_kernel(const KParams p, const long long ntiles, const TileKGeom G) {
    constexpr int NO = ndofs(DIM, ORDER), NE = NO * (NO + 1) / 2, NA = MOM ? mom_count<DIM>(2 * ORDER) : NE;
    constexpr int NRED = NA + NO, TC = KW;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sX = lds;
    double* sF = lds + TC * G.RS;
    double* sMax = lds + G.lds_main;

    const int tid = threadIdx.x, lane = tid & (KW - 1), wave = tid / KW;
    const int k0 = wave * G.KPW;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long j0 = tile * TC, j = j0 + lane;
        const bool valid = j < p.ncases;
        const long long jc = valid ? j : p.ncases - 1;
        const long long nvalid = (p.ncases - j0 < TC) ? (p.ncases - j0) : TC;

        const int nkc = min(p.nk[jc * p.snk], G.K);
        const bool uniform = (p.wm[jc * p.swm] == WLSQM_WEIGHT_UNIFORM);
        unsigned long long known, dropped;
        effective_mask<NO>(p.knowns[jc * p.sknowns], known, dropped);
        double xi[DIM];
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.xi[jc * p.sxi_j + m];

        // ---- staging in rounds of KROUND chunks per lane: xk block, then fk block
        {
            const kd2_* gx = reinterpret_cast<const kd2_*>(p.xk + j0 * (long long)(G.K * DIM));
            const long long xlim = nvalid * G.CPRX;
            for (int q0 = tid; q0 < G.XCH; q0 += KNT * KROUND) {
                kd2_ b[KROUND];
#pragma unroll
                for (int i = 0; i < KROUND; ++i) { const long long q = q0 + i * KNT; b[i] = gx[q < xlim ? q : xlim - 1]; }
#pragma unroll
                for (int i = 0; i < KROUND; ++i) {
                    const int q = q0 + i * KNT;
                    if (q < G.XCH) {
                        const int r = q / G.CPRX, c2 = q - r * G.CPRX;
                        double* d = sX + r * G.RS + 2 * c2;
                        d[0] = b[i].x; d[1] = b[i].y;
                    }
                }
            }
            const kd2_* gf = reinterpret_cast<const kd2_*>(p.fk + j0 * (long long)G.K);
            const long long flim = nvalid * G.CPRF;
            for (int q0 = tid; q0 < G.FCH; q0 += KNT * KROUND) {
                kd2_ b[KROUND];
#pragma unroll
                for (int i = 0; i < KROUND; ++i) { const long long q = q0 + i * KNT; b[i] = gf[q < flim ? q : flim - 1]; }
#pragma unroll
                for (int i = 0; i < KROUND; ++i) {
                    const int q = q0 + i * KNT;
                    if (q < G.FCH) {
                        const int r = q / G.CPRF, c2 = q - r * G.CPRF;
                        double* d = sF + r * G.FS + 2 * c2;
                        d[0] = b[i].x; d[1] = b[i].y;
                    }
                }
            }
        }
        __syncthreads();

        const double* xr = sX + lane * G.RS;
        const double* fr = sF + lane * G.FS;
        const int k1 = min(k0 + G.KPW, nkc);

        double max_d2 = 0.0;
        for (int k = k0; k < k1; ++k) {
            double d2 = 0.0;
#pragma unroll
            for (int m = 0; m < DIM; ++m) { const double dd = xr[k * DIM + m] - xi[m]; d2 += dd * dd; }
            max_d2 = d2 > max_d2 ? d2 : max_d2;
        }
        sMax[wave * TC + lane] = max_d2;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KSP; ++s) { const double o = sMax[s * TC + lane]; max_d2 = o > max_d2 ? o : max_d2; }
        const double inv_max = inverse_max(max_d2);

        double A[NA], g[NO];               // MOM: moments mu / nu; else the packed upper triangle of M / g
#pragma unroll
        for (int e = 0; e < NA; ++e) A[e] = 0.0;
#pragma unroll
        for (int a = 0; a < NO; ++a) g[a] = 0.0;
#pragma unroll 2
        for (int k = k0; k < k1; ++k) {
            double d[DIM];
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = xr[k * DIM + m] - xi[m];
            if constexpr (MOM) {
                double d2 = 0.0;
#pragma unroll
                for (int m = 0; m < DIM; ++m) d2 += d[m] * d[m];
                accumulate_moments_best<DIM, ORDER>(A, g, d, weight(d2, inv_max, uniform), fr[k]);
            } else {
                double cc[NO];
                const double d2 = monomials<DIM, ORDER>(d, cc);
                accumulate<NO>(A, g, cc, weight(d2, inv_max, uniform), fr[k]);
            }
        }

        __syncthreads();
        double* red = lds;
        if (wave > 0) {
            double* mine = red + (wave - 1) * (NRED * TC) + lane;
#pragma unroll
            for (int e = 0; e < NA; ++e) mine[e * TC] = A[e];
#pragma unroll
            for (int a = 0; a < NO; ++a) mine[(NA + a) * TC] = g[a];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 1; w < KSP; ++w) {
                const double* other = red + (w - 1) * (NRED * TC) + lane;
#pragma unroll
                for (int e = 0; e < NA; ++e) A[e] += other[e * TC];
#pragma unroll
                for (int a = 0; a < NO; ++a) g[a] += other[(NA + a) * TC];
            }
            constexpr unsigned long long FULL = (1ull << NO) - 1ull;
            if (valid && known != FULL) {
                double* fio = p.fi + j * p.sfi_j;
                auto finish = [&](double (&M)[NE], double (&rhs)[NO]) {
                    if (known) {
                        double val[NO];
#pragma unroll
                        for (int a = 0; a < NO; ++a) val[a] = (((known & ~dropped) >> a) & 1ull) ? fio[a] : 0.0;
                        eliminate_knowns<NO>(M, rhs, known, val);
                    }
                    ldlt_factor<NO>(M);
                    ldlt_solve<NO>(M, rhs);
#pragma unroll
                    for (int a = 0; a < NO; ++a)
                        if (!((known >> a) & 1ull)) fio[a] = rhs[a];
                };
                if constexpr (MOM) {
                    double M[NE], rhs[NO];
                    expand_moments<DIM, ORDER>(A, g, M, rhs);
                    finish(M, rhs);
                } else {
                    finish(A, g);
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One-wave variant (K <= 64): a 64-lane workgroup owns a tile of 16 cases, 4 lanes per case, lane (c, h) takes the
// neighbours [h*KPL, (h+1)*KPL).  Only xk goes through LDS; every lane reads its own fk values straight from global
// memory into registers (they are consumed late, after the distance pass, so their latency is hidden), the four lanes
// of a case meet through wave shuffles, and nothing needs a barrier between waves.  Measured on the benchmark shapes
// (fit_tile.hip) this shape beats four waves per 64-case tile by 7 % (C2) and 38 % (C5).
// FMAX: compile-time bound of the neighbours per lane (8 for K <= 32, 16 for K <= 64).  Every iteration sits under its own
//   wave-uniform `kk < KPL` branch; grouping four iterations per branch (straight-line code for the scheduler) was measured
//   and rejected: the extra scheduling freedom costs 40 VGPRs and a wave per SIMD (C2 at K = 32: 0.247 vs 0.221 ms).
// EXTRAS: sensitivities (impl.pyx:776-778, 821-846) and iterative refinement (impl.pyx:986-1083) on the same tile.  After
//   the shuffle butterfly the four lanes of a case hold bit-identical sums (fp addition commutes), so each of them
//   factors the matrix for itself at no extra cost (the instructions run for the whole wave anyway) and then handles
//   ITS neighbours: one substitution per neighbour for the sensitivities, its share of the residual and of the
//   correction's right-hand side per refinement sweep (met again through the butterfly).
//   A K-specialised instantiation of this kernel (geometry as compile-time constants, predicates folded) does not help the
//   extras: C2 at K = 32, iterative 1.08 instead of 1.10 ms, do_sens 1.80 instead of 0.73 ms (the unrolled substitutions
//   spill) — unlike the basic fit, where fixed K is worth 20-70 % (fit_tile.hip).
//   The 3D order-2 / 2D order-3 instantiations (55-entry factor) spill 0.4-0.9 KB per lane; parking the factor in LDS (one
//   copy per case, substitutions read it from there) made it worse — the compiler hoists the LDS reads of the unrolled
//   substitutions back into registers: 0.75-1.2 KB of spills, C5 do_sens 4.68 instead of 3.72 ms, iterative 4.06 instead of
//   3.18 ms.  (The generic kernel does C5's do_sens in 3.58 ms and its iterative fit in 4.62 ms.)
//   A 15-unknown instantiation (2D order 4) fills all 512 registers of a lone wave and still spills 1.1 KB: 400k C3 cases,
//   do_sens 3.96 and iterative 2.54 ms against 2.53 / 2.35 ms on the generic kernel, which keeps those.
// LPC: lanes per case (4 on 16-case tiles; 2 on 32-case tiles, which halves the butterflies and the redundant factorisations
//   per case and wins for K <= 32 under the oversubscribed grids, like the fixed-K shapes of fit_tile_even.hip).
// MINW: waves per SIMD the register allocator plans for.  1 lets the heavy extras instantiations keep everything in the
//   512 registers of a lone wave (no scratch) instead of spilling at 2.
template <int DIM, int ORDER, bool MOM, int FMAX, bool EXTRAS = false, int LPC = K1_LPC, int MINW = 2>
__global__ __launch_bounds__(KW, MINW) void fit_tile1_kernel(const KParams p, const long long ntiles, const Tile1Geom G) {
    constexpr int NO = ndofs(DIM, ORDER), NE = NO * (NO + 1) / 2, NA = MOM ? mom_count<DIM>(2 * ORDER) : NE, TC = KW / LPC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sX = lds;
    const int lane = threadIdx.x, c = lane % TC, h = lane / TC;
    const int k0 = h * G.KPL;

    const long long ncases = live_cases(p);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // position in the launch -> case number: all cases in order, or the index list of an order bucket (p.case_index)
        const long long j0 = tile * TC, pos = j0 + c;
        if (j0 >= ncases) break;                                  // (a bucket sized on the device may be shorter than the launch)
        const bool valid = pos < ncases;
        const long long posc = valid ? pos : ncases - 1;
        const long long jc = p.case_index ? p.case_index[posc] : posc, j = jc;
        const long long nvalid = (ncases - j0 < TC) ? (ncases - j0) : TC;

        const int nkc = min(p.nk[jc * p.snk], G.K);
        const bool uniform = (p.wm[jc * p.swm] == WLSQM_WEIGHT_UNIFORM);
        unsigned long long known, dropped;
        effective_mask<NO>(p.knowns[jc * p.sknowns], known, dropped);
        double xi[DIM];
        double fdir[FMAX];
        if (p.hoods) {
            // ---- index-based input: this lane gathers ITS neighbours' rows of the point tables S / F through hoods[j, k]
            // and parks the coordinates in the same padded LDS image the dense path builds (its own slots of row c)
            const long long pj = own_point(p, jc);
#pragma unroll
            for (int m = 0; m < DIM; ++m) xi[m] = p.S[pj * DIM + m];
            const int* hr = p.hoods + jc * p.shoods_j;
            double* row = sX + c * G.RS;
#pragma unroll
            for (int kk = 0; kk < FMAX; ++kk)
                if (kk < G.KPL) {
                    const int k = k0 + kk;
                    // slots k >= nk[j] are padding and may hold anything (-1, npoints, ...): never dereferenced
                    const long long idx = k < nkc ? (long long)hr[k] : pj;
                    fdir[kk] = p.F[idx];
                    if (k < G.K) {
#pragma unroll
                        for (int m = 0; m < DIM; ++m) row[k * DIM + m] = p.S[idx * DIM + m];
                    }
                }
        } else {
#pragma unroll
            for (int m = 0; m < DIM; ++m) xi[m] = p.xi[jc * p.sxi_j + m];
            // ---- this lane's fk values (clamped inside the row; slots beyond nk[j] are masked below)
            tile1_load_f<FMAX>(fdir, p.fk + jc * (long long)G.K, k0, G);
            // ---- the tile's xk block: coalesced 16 B per lane, K1_ROUND loads in flight, parked in padded LDS rows
            if (p.case_index) tile1_stage_x_indexed<DIM>(sX, p.xk, jc, nvalid, lane, G);
            else tile1_stage_x<DIM>(sX, p.xk + j0 * (long long)(G.K * DIM), nvalid, lane, G);
        }
        __syncthreads();

        const double* xr = sX + c * G.RS;
        const double max_d2 = tile1_max_d2<DIM, FMAX, LPC>(xr, xi, k0, nkc, G);
        const double inv_max = inverse_max(max_d2);

        double A[NA], g[NO];               // MOM: moments mu / nu; else the packed upper triangle of M / g
#pragma unroll
        for (int e = 0; e < NA; ++e) A[e] = 0.0;
#pragma unroll
        for (int a = 0; a < NO; ++a) g[a] = 0.0;
#pragma unroll
        for (int kk = 0; kk < FMAX; ++kk) {
            if (kk < G.KPL) {              // wave-uniform
                const int k = k0 + kk;
                const bool live = k < nkc;
                const int kc = live ? k : 0;
                double d[DIM];
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = live ? xr[kc * DIM + m] - xi[m] : 0.0;
                const double f = live ? fdir[kk] : 0.0;
                if constexpr (MOM) {
                    double d2 = 0.0;
#pragma unroll
                    for (int m = 0; m < DIM; ++m) d2 += d[m] * d[m];
                    accumulate_moments_best<DIM, ORDER>(A, g, d, live ? weight(d2, inv_max, uniform) : 0.0, f);
                } else {
                    double cc[NO];
                    const double d2 = monomials<DIM, ORDER>(d, cc);
                    accumulate<NO>(A, g, cc, live ? weight(d2, inv_max, uniform) : 0.0, f);
                }
            }
        }
        // ---- the four lanes of a case
#pragma unroll
        for (int off = TC; off < KW; off <<= 1) {
#pragma unroll
            for (int e = 0; e < NA; ++e) A[e] += __shfl_xor(A[e], off, KW);
#pragma unroll
            for (int a = 0; a < NO; ++a) g[a] += __shfl_xor(g[a], off, KW);
        }
        constexpr unsigned long long FULL = (1ull << NO) - 1ull;
        if constexpr (!EXTRAS) {
            if (valid && h == 0 && known != FULL) {
                double* fio = p.fi + j * p.sfi_j;
                auto finish = [&](double (&M)[NE], double (&rhs)[NO]) {
                    if (known) {
                        double val[NO];
#pragma unroll
                        for (int a = 0; a < NO; ++a) val[a] = (((known & ~dropped) >> a) & 1ull) ? fio[a] : 0.0;
                        eliminate_knowns<NO>(M, rhs, known, val);
                    }
                    ldlt_factor<NO>(M);
                    ldlt_solve<NO>(M, rhs);
#pragma unroll
                    for (int a = 0; a < NO; ++a)
                        if (!((known >> a) & 1ull)) fio[a] = rhs[a];
                };
                if constexpr (MOM) {
                    double M[NE], rhs[NO];
                    expand_moments<DIM, ORDER>(A, g, M, rhs);
                    finish(M, rhs);
                } else {
                    finish(A, g);
                }
            }
        } else {
            // every lane factors its case (all four lanes of a case: same bits) and keeps the factor for the extras
            double* fio = p.fi + jc * p.sfi_j;
            const bool store = valid && known != FULL;             // nr < 1: no-op (impl.pyx:574, 636, 742)
            double M[NE], sol[NO], val[NO];
            if constexpr (MOM) expand_moments<DIM, ORDER>(A, g, M, sol);
            else {
#pragma unroll
                for (int e = 0; e < NE; ++e) M[e] = A[e];
#pragma unroll
                for (int a = 0; a < NO; ++a) sol[a] = g[a];
            }
#pragma unroll
            for (int a = 0; a < NO; ++a) val[a] = (((known & ~dropped) >> a) & 1ull) ? fio[a] : 0.0;
            eliminate_knowns<NO>(M, sol, known, val);
            ldlt_factor<NO>(M);
            ldlt_solve<NO>(M, sol);

            // monomials and weight of this lane's neighbour slot kk (rows are still in LDS)
            auto neighbour = [&](int kk, double (&cc)[NO], double& w) {
                const int k = k0 + kk;
                const int kc = k < nkc ? k : 0;
                double d[DIM];
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = xr[kc * DIM + m] - xi[m];
                const double d2 = monomials<DIM, ORDER>(d, cc);
                w = weight(d2, inv_max, uniform);
                return k < nkc;
            };

            // ---- sensitivities: sens[k, a] = d fi[a] / d fk[k]; NaN for knowns (impl.pyx:821-823)
            if (p.do_sens && p.sens) {
                double* sr = p.sens + jc * p.ss_j;
                auto sens_row = [&](int k, double (&sv)[NO]) {        // the row of neighbour k (clamped) of this lane's case
                    const int kc = k < nkc ? k : 0;
                    double d[DIM], cc[NO];
#pragma unroll
                    for (int m = 0; m < DIM; ++m) d[m] = xr[kc * DIM + m] - xi[m];
                    const double d2 = monomials<DIM, ORDER>(d, cc);
                    const double w = weight(d2, inv_max, uniform);
#pragma unroll
                    for (int a = 0; a < NO; ++a) sv[a] = ((known >> a) & 1ull) ? 0.0 : ((a == 0) ? w : w * cc[a]);
                    ldlt_solve<NO>(M, sv);
                };
                // dense output (rows of exactly `no` doubles, K rows per case): a case's rows are one contiguous run, so
                // the tile's rows go through LDS in slabs of 8 consecutive neighbours per case (lane (c, h) takes
                // neighbour 4*kk + h here) and leave as full-line stores of 8 B per lane.  Direct 8-byte stores from
                // the owner lanes touch 64 lines per instruction and cost 0.73 ms per 1M C2 cases against 0.45 ms for
                // everything else in this kernel.
                const bool dense = p.ss_k == NO && p.ss_j == (long long)G.K * NO && !p.case_index && !__any(dropped != 0) &&
                                   (reinterpret_cast<uintptr_t>(p.sens) & 15u) == 0;
                if (dense) {
                    constexpr int SL = 2, E = SL * LPC * NO, CS = E + 1;      // elements per case and slab; padded stride
                    double* sS = lds + TC * G.RS;                                // [TC][CS]
                    int* sNk = reinterpret_cast<int*>(sS + TC * CS);             // rows each case stores (0: none)
                    if (h == 0) sNk[c] = store ? nkc : 0;
                    const int nslab = (G.KPL + SL - 1) / SL;
                    double* out0 = p.sens + j0 * p.ss_j;
                    for (int slab = 0; slab < nslab; ++slab) {
#pragma unroll
                        for (int i = 0; i < SL; ++i) {
                            double sv[NO];
                            sens_row((slab * SL + i) * LPC + h, sv);
                            double* q = sS + c * CS + (i * LPC + h) * NO;
#pragma unroll
                            for (int a = 0; a < NO; ++a) q[a] = ((known >> a) & 1ull) ? __longlong_as_double(0x7ff8000000000000LL) : sv[a];
                        }
                        __syncthreads();
                        const int kbase = slab * SL * LPC;
                        if constexpr (NO % 2 == 0) {
                            // 16 B per lane: a pair never straddles two neighbours (NO even) nor two cases (E even)
#pragma unroll
                            for (int q0 = 0; q0 < TC * E / 2; q0 += KW) {
                                const int q = q0 + lane;
                                if ((TC * E / 2) % KW == 0 || q < TC * E / 2) {
                                    const int cs = q / (E / 2), e = 2 * (q - cs * (E / 2)), k = kbase + e / NO;
                                    if (k < sNk[cs]) {
                                        k1d2_ v; v.x = sS[cs * CS + e]; v.y = sS[cs * CS + e + 1];
#if WLSQM_TILEK_SENS_NT
                                        __builtin_nontemporal_store(v, reinterpret_cast<k1d2_*>(out0 + (long long)cs * p.ss_j + (long long)kbase * NO + e));
#else
                                        *reinterpret_cast<k1d2_*>(out0 + (long long)cs * p.ss_j + (long long)kbase * NO + e) = v;
#endif
                                    }
                                }
                            }
                        } else {
#pragma unroll
                        for (int q0 = 0; q0 < TC * E; q0 += KW) {
                            const int q = q0 + lane;
                            if ((TC * E) % KW == 0 || q < TC * E) {
                                const int cs = q / E, e = q - cs * E, k = kbase + e / NO;       // compile-time divisors
#if WLSQM_TILEK_SENS_NT
                                if (k < sNk[cs]) __builtin_nontemporal_store(sS[cs * CS + e], &out0[(long long)cs * p.ss_j + (long long)kbase * NO + e]);
#else
                                if (k < sNk[cs]) out0[(long long)cs * p.ss_j + (long long)kbase * NO + e] = sS[cs * CS + e];
#endif
                            }
                        }
                        }
                        __syncthreads();
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < FMAX; ++kk) {
                        if (kk < G.KPL) {              // wave-uniform
                            double sv[NO];
                            sens_row(k0 + kk, sv);
                            if (k0 + kk < nkc && store) {
                                double* row = sr + (long long)(k0 + kk) * p.ss_k;
#pragma unroll
                                for (int a = 0; a < NO; ++a) {
                                    if (!((known >> a) & 1ull)) row[a] = sv[a];
                                    else if (!((dropped >> a) & 1ull)) row[a] = __longlong_as_double(0x7ff8000000000000LL);
                                }
                            }
                        }
                    }
                }
            }

            // ---- iterative refinement: every sweep re-evaluates the model at the neighbours, solves for a correction
            // from the residual and stops when the residual's max-norm repeats exactly (impl.pyx:1037-1057)
            int iters = 0;
            bool unfinished = false;
            if (p.iterative) {
                double fi[NO];
#pragma unroll
                for (int a = 0; a < NO; ++a) fi[a] = ((known >> a) & 1ull) ? val[a] : sol[a];
#pragma unroll
                for (int a = 0; a < NO; ++a) if ((dropped >> a) & 1ull) fi[a] = fio[a];   // Case_set_fi copies all `no`
                double prev_norm = -1.0;
                // a later ROUND of the refinement (KParams::it_first): the iterate so far is in fi, the last residual norm in it_state
                const int i_first = p.it_stop > 0 ? p.it_first : 0, i_stop = p.it_stop > 0 ? min(p.it_stop, p.max_iter) : p.max_iter;
                if (i_first > 0) {
#pragma unroll
                    for (int a = 0; a < NO; ++a) fi[a] = fio[a];
                    prev_norm = p.it_state[jc];
                }
                bool running = true, broke = false;
                int i = 0;
                // the weights of this lane's neighbours do not change from sweep to sweep (a quotient, a root): kept
                double wv[FMAX];
#pragma unroll
                for (int kk = 0; kk < FMAX; ++kk) {
                    wv[kk] = 0.0;
                    if (kk < G.KPL) { double cc0[NO], w0; const bool live0 = neighbour(kk, cc0, w0); wv[kk] = live0 ? w0 : 0.0; }
                }
                for (i = i_first; i < i_stop; ++i) {
                    if (!__any(running)) break;
                    double norm = 0.0, r[NO];
#pragma unroll
                    for (int a = 0; a < NO; ++a) r[a] = 0.0;
#pragma unroll
                    for (int kk = 0; kk < FMAX; ++kk) {
                        if (kk < G.KPL) {
                            double cc[NO];
                            const int k = k0 + kk;
                            const bool live = k < nkc;
                            {
                                const int kc = live ? k : 0;
                                double d[DIM];
#pragma unroll
                                for (int m = 0; m < DIM; ++m) d[m] = xr[kc * DIM + m] - xi[m];
                                (void)monomials<DIM, ORDER>(d, cc);
                            }
                            const double w = wv[kk];
                            double model = fi[0];
#pragma unroll
                            for (int a = 1; a < NO; ++a) model += cc[a] * fi[a];
                            const double res = live ? fdir[kk] - model : 0.0;
                            const double ar = fabs(res);
                            norm = ar > norm ? ar : norm;
                            const double wr = w * res;
#pragma unroll
                            for (int a = 0; a < NO; ++a) r[a] += (a == 0) ? wr : wr * cc[a];
                        }
                    }
#pragma unroll
                    for (int off = TC; off < KW; off <<= 1) {
                        const double o = __shfl_xor(norm, off, KW);
                        norm = o > norm ? o : norm;
#pragma unroll
                        for (int a = 0; a < NO; ++a) r[a] += __shfl_xor(r[a], off, KW);
                    }
                    if (running) {
                        if (norm == prev_norm) { running = false; broke = true; iters = i; }
                        else {
                            prev_norm = norm;
#pragma unroll
                            for (int a = 0; a < NO; ++a) if ((known >> a) & 1ull) r[a] = 0.0;
                            ldlt_solve<NO>(M, r);
#pragma unroll
                            for (int a = 0; a < NO; ++a) if (!((known >> a) & 1ull)) fi[a] += r[a];
                        }
                    }
                }
                if (!broke) iters = p.max_iter > 0 ? p.max_iter : 1;   // for/else, impl.pyx:1080-1081
#pragma unroll
                for (int a = 0; a < NO; ++a) sol[a] = fi[a];
                // the round is over and this case is still running: the next round carries on from the iterate stored below
                unfinished = running && !broke && i_stop < p.max_iter;
                if (unfinished && store && h == 0 && p.cont_list) {
                    const unsigned long long slot = atomicAdd(reinterpret_cast<unsigned long long*>(p.cont_count), 1ull);
                    p.cont_list[slot] = jc;
                    p.it_state[jc] = prev_norm;
                }
            }
            if (store && h == 0) {
#pragma unroll
                for (int a = 0; a < NO; ++a)
                    if (!((known >> a) & 1ull)) fio[a] = sol[a];
                if (p.iterative && p.iters_out && !unfinished) atomicMax(p.iters_out, iters);
            }
        }
        __syncthreads();   // the next tile overwrites LDS
    }
}

__global__ void tile1_rounds_zero_kernel(long long* ws) { ws[threadIdx.x] = 0; }

static int rup(int v, int m, int r) { return v + ((r - v % m) % m + m) % m; }

template <int DIM, int ORDER, bool MOM>
static int launch_tilek(const KParams& p, long long K, hipStream_t stream, bool* handled) {
    constexpr int NO = ndofs(DIM, ORDER), NE = NO * (NO + 1) / 2, NRED = (MOM ? mom_count<DIM>(2 * ORDER) : NE) + NO;
    TileKGeom G;
    G.K = (int)K; G.KPW = (int)((K + KSP - 1) / KSP);
    G.RS = rup((int)K * DIM, 2, 1); G.FS = rup((int)K, 2, 1);       // odd strides: conflict-free ds_read_b64
    G.XCH = KW * (int)K * DIM / 2; G.FCH = KW * (int)K / 2;
    G.CPRX = (int)K * DIM / 2; G.CPRF = (int)K / 2;
    const int lds_tile = KW * (G.RS + G.FS), lds_red = (KSP - 1) * NRED * KW;
    G.lds_main = lds_tile > lds_red ? lds_tile : lds_red;
    const size_t lds_bytes = sizeof(double) * (size_t)(G.lds_main + KSP * KW);
    if (lds_bytes > 80 * 1024) return WLSQM_OK;                     // keep >= 2 workgroups per CU; larger K: generic kernel
    *handled = true;
    const long long ntiles = (p.ncases + KW - 1) / KW;
    auto kern = fit_tilek_kernel<DIM, ORDER, MOM>;
    static KernelSetup setup;
    long long grid = 0;
    int rc = persistent_grid(reinterpret_cast<const void*>(kern), KNT, lds_bytes, 80 * 1024, false, setup, &grid);
    if (rc != WLSQM_OK) return rc;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(KNT), lds_bytes, stream, p, ntiles, G);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("tilek");
    return WLSQM_OK;
}

template <int DIM, int ORDER, bool MOM, int FMAX, bool EXTRAS = false, int LPC = K1_LPC, int MINW = 2>
static int launch_tile1(const KParams& p, long long K, hipStream_t stream, bool* handled) {
    Tile1Geom G;
    if (!tile1_geometry<DIM, LPC>(K, FMAX, G)) return WLSQM_OK;
    constexpr int K1_TC = K1_WV / LPC;          // (shadows the default tile size below)
    constexpr int NO_ = ndofs(DIM, ORDER);
    // EXTRAS: + the sensitivities' staging slab [TC][2*4*no + 1] and 16 ints
    const size_t lds_bytes = sizeof(double) * (size_t)(K1_TC * G.RS + (EXTRAS ? K1_TC * (2 * LPC * NO_ + 1) + K1_TC / 2 : 0));
    *handled = true;
    const long long ntiles = (p.ncases + K1_TC - 1) / K1_TC;
    auto kern = fit_tile1_kernel<DIM, ORDER, MOM, FMAX, EXTRAS, LPC, MINW>;
    static KernelSetup setup;
    long long grid = 0;
    int rc = persistent_grid(reinterpret_cast<const void*>(kern), KW, lds_bytes, 0, false, setup, &grid);
    if (rc != WLSQM_OK) return rc;
    if (grid > ntiles) grid = ntiles;
    if constexpr (EXTRAS) {
        // Refinement in ROUNDS (KParams::it_first): sweeps [0, 3) for every case, [3, 6) for the cases still running, the rest for
        // those that still are — each round a launch over the previous round's survivor list, sized on the device (no host
        // synchronisation); bit-identical to the single launch (a case's arithmetic involves its own lanes only; tools/time_rounds.py).
        // OFF by default (WLSQM_HIP_REFINE_ROUNDS=1 turns it on): measured SLOWER, 1.75 against 0.88 ms per 1M configs[1] cases and
        // 3.33 against 2.34 ms on configs[4].  The reference-order arithmetic's stop test fires after 3.1 / 3.4 sweeps on average there (1 024 cases at 1M
        // density: 33 / 507 / 233 / 110 / 45 / 27 / 16 / 7 / 4 / 42 cases for 1 .. 10), which is what made the rounds look worthwhile
        // — but the test is exact equality of two consecutive residual max-norms, and in THIS kernel's arithmetic (moment form, FMA
        // contraction, LDL^T) the iterates keep moving in their last bits for longer: 67-76 % of the cases are still running after
        // sweep 3 and 24-38 % after sweep 6 (WLSQM_HIP_REFINE_DEBUG=1), and a later round has to rebuild the factor of its cases
        // (configs[4]: 1.0 ms of the kernel's 1.45 ms for fit + 3 sweeps).  The divergence inside a tile is therefore small on the
        // GPU; what the refinement lines cost is the sweeps themselves.
        const char* e = getenv("WLSQM_HIP_REFINE_ROUNDS");
        const bool rounds = p.iterative && !p.do_sens && p.it_stop == 0 && p.max_iter >= 5 && p.ncases >= 4096 && (e && e[0] == '1');
        if (rounds) {
            const long long n = p.ncases;
            long long* ws = nullptr;                  // [8] counters, [n] list A, [n] list B, [n] residual norms
            rc = scratch_alloc_async(reinterpret_cast<void**>(&ws), (size_t)(3 * n + 8) * sizeof(long long), stream);
            if (rc != WLSQM_OK) return rc;
            hipLaunchKernelGGL(tile1_rounds_zero_kernel, dim3(1), dim3(8), 0, stream, ws);
            const int r1 = 3, r2 = p.max_iter > 7 ? 6 : p.max_iter;
            KParams q = p;
            q.it_first = 0; q.it_stop = r1; q.cont_list = ws + 8; q.cont_count = ws; q.it_state = reinterpret_cast<double*>(ws + 8 + 2 * n);
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(KW), lds_bytes, stream, q, ntiles, G);
            q.case_index = ws + 8; q.ncases_dev = ws; q.ncases = n;
            q.it_first = r1; q.it_stop = r2; q.cont_list = ws + 8 + n; q.cont_count = ws + 1;
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(KW), lds_bytes, stream, q, ntiles, G);
            if (r2 < p.max_iter) {
                q.case_index = ws + 8 + n; q.ncases_dev = ws + 1;
                q.it_first = r2; q.it_stop = p.max_iter; q.cont_list = nullptr; q.cont_count = nullptr;
                hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(KW), lds_bytes, stream, q, ntiles, G);
            }
            hipError_t le = hipGetLastError();
            if (getenv("WLSQM_HIP_REFINE_DEBUG")) {           // survivors of the first two rounds (synchronises: diagnostics only)
                long long c[2] = {0, 0};
                (void)hipMemcpyAsync(c, ws, sizeof(c), hipMemcpyDeviceToHost, stream);
                (void)hipStreamSynchronize(stream);
                fprintf(stderr, "refinement rounds: %lld cases, %lld still running after sweep %d, %lld after sweep %d\n", n, c[0], r1, c[1], r2);
            }
            const int rf = scratch_free_async(ws, stream);
            if (le != hipSuccess) return hip_fail(le, "fit_tile1_kernel (refinement rounds)");
            note_kernel("tile1-extras");
            return rf;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(KW), lds_bytes, stream, p, ntiles, G);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel(EXTRAS ? "tile1-extras" : "tile1");
    return WLSQM_OK;
}

// Runtime-K tile path: dense contiguous arrays with 16-byte rows (K*dim and K even), no extras, no bucketing.
int launch_fit_tilek(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* off = getenv("WLSQM_HIP_DISABLE_TILE");
    if (off && off[0] == '1') return WLSQM_OK;
    if (p.hoods) {                                               // index-based input: the one-wave kernel gathers per lane
        if (K < 4 || p.shoods_j != K) return WLSQM_OK;
    } else {
        if (K < 4 || ((K * dimension) % 2) != 0) return WLSQM_OK;   // rows of xk are multiples of 16 bytes
        if (p.sxk_k != dimension || p.sxk_j != K * dimension || p.sfk_k != 1 || p.sfk_j != K) return WLSQM_OK;
        if ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(p.fk)) & 15u) return WLSQM_OK;
    }
    // Two shapes (A/B over K = 16..64, tools/tune.py w1/w4): one wave per 16-case tile wins for 2D order 2 and 3D order 1
    // (+5..+20 %) and is the only one whose LDS image fits for large K; four waves per 64-case tile wins for the
    // register-heavy systems (3D order 2, 2D order 3: the one-wave kernel spills there) and for order <= 1 with few
    // neighbours.  WLSQM_TILEK_SHAPE=1|4 forces a shape (A/B).
    const bool extras = p.do_sens || p.iterative;
    if (extras) {
        // sensitivities / iterative refinement: the one-wave kernel with EXTRAS (K <= 64), else the generic kernel
        const char* ex = getenv("WLSQM_HIP_DISABLE_TILE_EXTRAS");
        if (ex && ex[0] == '1') return WLSQM_OK;
        // `lone`: instantiations whose registers are planned for ONE wave per SIMD (__launch_bounds__(64, 1): up to 512 registers,
        // no scratch) instead of two.  The 10-unknown systems (2D order 3, 3D order 2) spill 0.4-1.2 KB per lane otherwise, and so
        // do the 32-neighbours-per-lane shapes.  400k cases, lone / paired, ms: 3D order 2 do_sens at K = 12 / 24 / 40 / 64 / 124:
        // 0.27 / 0.51 / 0.73 / 1.03 / 3.58 against 0.38 / 0.75 / 1.59 / 2.63 / 7.60 (generic kernel at K = 40 / 124: 1.43 / 5.6), its
        // iterative fit 0.41 / 0.81 / 1.08 / 1.37 / 4.94 against 0.32 / 0.60 / 1.41 / 2.24 / 8.18; 2D order 3 do_sens at K = 16 / 32 /
        // 48: 0.29 / 0.55 / 0.77 against 0.35 / 0.74 / 1.48, iterative 0.46 / 0.87 / 1.12 against 0.37 / 0.69 / 1.05; 2D order 2 at K =
        // 48 / 64: 0.54 / 0.66 against 0.44 / 0.57 (paired stays), at K = 80 / 124: 0.84 / 1.18 against 1.48 / 2.65.
        const bool heavy = wlsqm_hip_number_of_dofs(dimension, order) > 6;
        const bool lone = heavy && (p.do_sens || K > 32);
#define XP(D, O, FM, LL) if (dimension == D && order == O) return launch_tile1<D, O, (O >= 2), FM, true, LL, 2>(p, K, stream, handled);
#define XL(D, O, FM, LL) if (dimension == D && order == O) return launch_tile1<D, O, (O >= 2), FM, true, LL, 1>(p, K, stream, handled);
#define XB(D, O, FM, LL)                                                                                           \
    if (dimension == D && order == O)                                                                              \
        return lone ? launch_tile1<D, O, (O >= 2), FM, true, LL, 1>(p, K, stream, handled)                         \
                    : launch_tile1<D, O, (O >= 2), FM, true, LL, 2>(p, K, stream, handled);
        if (K > K1_LPC * K1_FMAX) {
            // 64 < K <= 128: 32 neighbours per lane, always a lone wave (2D orders 1-2, 3D orders 1-2)
            if (K > K1_LPC * 32) return WLSQM_OK;
            XL(2, 1, 32, K1_LPC) XL(2, 2, 32, K1_LPC) XL(3, 1, 32, K1_LPC) XL(3, 2, 32, K1_LPC)
            return WLSQM_OK;
        }
        if (K <= 2 * 8) {
            // up to 16 neighbours: two lanes per case on 32-case tiles.  1M cases, 2D order 2 at K = 8 / 12 / 16: do_sens 0.225 /
            // 0.276 / 0.336 ms and iterative 0.418 / 0.485 / 0.563 ms against 0.322 / 0.397 / 0.429 and 0.867 / 0.787 / 0.788 with
            // four lanes.  (With 16 neighbours per lane, K <= 32, the two-lane shape loses: C2 0.73 / 1.10 against 0.66 / 1.00 ms.)
            XP(1, 0, 8, 2) XP(1, 1, 8, 2) XP(1, 2, 8, 2) XP(1, 3, 8, 2) XP(1, 4, 8, 2)
            XP(2, 0, 8, 2) XP(2, 1, 8, 2) XP(2, 2, 8, 2) XB(2, 3, 8, 2)
            XP(3, 0, 8, 2) XP(3, 1, 8, 2) XB(3, 2, 8, 2)
        }
        if (K <= 32) {
            XP(1, 0, 8, K1_LPC) XP(1, 1, 8, K1_LPC) XP(1, 2, 8, K1_LPC) XP(1, 3, 8, K1_LPC) XP(1, 4, 8, K1_LPC)
            XP(2, 0, 8, K1_LPC) XP(2, 1, 8, K1_LPC) XP(2, 2, 8, K1_LPC) XB(2, 3, 8, K1_LPC)
            XP(3, 0, 8, K1_LPC) XP(3, 1, 8, K1_LPC) XB(3, 2, 8, K1_LPC)
        }
        XP(1, 0, K1_FMAX, K1_LPC) XP(1, 1, K1_FMAX, K1_LPC) XP(1, 2, K1_FMAX, K1_LPC) XP(1, 3, K1_FMAX, K1_LPC) XP(1, 4, K1_FMAX, K1_LPC)
        XP(2, 0, K1_FMAX, K1_LPC) XP(2, 1, K1_FMAX, K1_LPC) XP(2, 2, K1_FMAX, K1_LPC) XL(2, 3, K1_FMAX, K1_LPC)
        XP(3, 0, K1_FMAX, K1_LPC) XP(3, 1, K1_FMAX, K1_LPC) XL(3, 2, K1_FMAX, K1_LPC)
#undef XP
#undef XL
#undef XB
        return WLSQM_OK;
    }
    const char* sv = getenv("WLSQM_TILEK_SHAPE");
    // the four-wave shape stages fk rows in 16-byte chunks and takes whole batches only (no index list)
    const bool can1 = K <= K1_LPC * K1_FMAX, can4 = (K % 2) == 0 && !p.case_index && !p.hoods;
    bool first1 = (dimension == 2 && order == 2) || (dimension == 3 && order == 1);
    if (sv && sv[0] == '1') first1 = true;
    if (sv && sv[0] == '4') first1 = false;
    // up to 32 neighbours: two lanes per case on 32-case tiles (1M cases, 2D order 2 at K = 15 / 16 / 31 / 32: 0.133 / 0.133 / 0.223 /
    // 0.219 ms against 0.196 / 0.192 / 0.249 / 0.247 with four lanes; 3D order 2 at K = 12: 0.176 against 0.289)
#define KCASE(D, O)                                                                            \
    if (dimension == D && order == O) {                                                        \
        for (int attempt = 0; attempt < 2 && !*handled; ++attempt) {                           \
            const bool one = (attempt == 0) == first1;                                         \
            int rc = WLSQM_OK;                                                                 \
            if (one && can1) rc = K <= 16 ? launch_tile1<D, O, (O >= 2), 8, false, 2>(p, K, stream, handled)  \
                                  : K <= 32 ? launch_tile1<D, O, (O >= 2), 16, false, 2>(p, K, stream, handled) \
                                          : launch_tile1<D, O, (O >= 2), 16>(p, K, stream, handled); \
            if (!one && can4) rc = launch_tilek<D, O, (O >= 2)>(p, K, stream, handled);        \
            if (rc != WLSQM_OK) return rc;                                                     \
        }                                                                                      \
        return WLSQM_OK;                                                                       \
    }
    KCASE(1, 0) KCASE(1, 1) KCASE(1, 2) KCASE(1, 3) KCASE(1, 4)
    KCASE(2, 0) KCASE(2, 1) KCASE(2, 2) KCASE(2, 3)
    KCASE(3, 0) KCASE(3, 1) KCASE(3, 2)
#undef KCASE
    return WLSQM_OK;
}

}  // namespace wlsqm
