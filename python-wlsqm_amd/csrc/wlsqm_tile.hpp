// wlsqm_tile.hpp — the fixed-K tile kernel (template) and its launcher, shared by the dispatch tables in fit_tile*.hip.
// Contiguous fast path of the WLSQM fit for gfx950: LDS-staged tiles.
//
// Same arithmetic as fit_lane.hip (the normal matrix of a case lives in VGPRs; see
// wlsqm_kernels.hpp for the reference citations), but the dense reference layout
// xk[ncases, K, dim], fk[ncases, K] is "array of structures" for a lane-per-case mapping, so
// the tile kernel moves it through LDS:
//
//   * a workgroup of KSPLIT waves owns a tile of TC = 64/LPC consecutive cases; the tile's xk
//     and fk blocks are single contiguous byte ranges in HBM and are read with fully coalesced
//     16-byte-per-lane loads, ALL issued before the first is consumed, then parked in LDS with
//     a padded row per case;
//   * lane (h, c) of every wave reads row c back with conflict-free ds_read_b128/b64 (row
//     stride chosen so that 16/32 consecutive lanes cover all 64 banks) and accumulates the
//     neighbours of share s = wave*LPC + h (k ascending inside a share);
//   * the LPC lanes of a case are summed with wave shuffles, the KSPLIT waves through LDS
//     (reusing the tile's storage); wave 0 does knowns elimination + LDL^T + substitution and
//     writes the `no` results.
//
// Shapes (KSPLIT waves x LPC lanes per case) are picked per configuration by A/B measurement, see launch_fit_tile.
// What the round-1 measurements say (1M cases, tools/tune.py):
//   * moment form (MOM, wlsqm_moments.hpp): fewer accumulators and operations per neighbour from order 2 up;
//   * one wave per 16-case tile (KSPLIT 1, LPC 4) with fk read straight from global memory (FKD) is the fastest dense
//     shape for 2D order 2 and 3D order 2: no barriers between waves, 8-16 KB of LDS per wave, and the fk loads are
//     consumed only after the distance pass.  Without FKD the same shape loses to four waves per 64-case tile;
//   * rejected for C2: software prefetch of the next tile through registers, early (-15 %) or late, during the solve
//     (-12 %): both cost a resident workgroup; expanding the moments from LDS in wave 0 (-5 %); bringing the NEXT tile's
//     xk in by LDS-DMA (global_load_lds_dwordx4 into a two-tile ring, rows rotated on the source side for conflict-free
//     reads, next tile's fk and scalars in a second register set; git history: fit_glds.hip): 0.193 vs 0.177 ms — the
//     second register set costs the third wave per SIMD, and at this occupancy the other waves already hide the wait;
//     non-temporal loads (__builtin_nontemporal_load) for the streamed-once xk / fk: -2..-3 % on C2, C5 and C3;
//     XCD-aware tile order for the index-based path (each XCD's workgroups stride through one contiguous eighth of the
//     tiles, so that the point rows shared by neighbouring tiles meet in one L2): +1 %, inside the noise — left out;
//     squeezing C5 into 168 VGPRs for a third wave per SIMD (solving lane parks its moments in LDS and expands from
//     there; unroll 1-5): 320 B of spills remain (45 accumulators + chain temporaries + 10 fk values) and the kernel
//     runs 2.4x slower; the same for C2 at 128 VGPRs (four waves): -15 %;
//     gfx950's v_permlane16_swap / v_permlane32_swap instead of the ds_bpermute butterflies (2 moves + 1 add per double
//     and step, no LDS crossbar): correct, but C5 0.46 instead of 0.36 ms and do_sens 0.82 instead of 0.70 ms (both
//     operands are overwritten, so every value needs two copies first); wave-shuffle instead of LDS for the per-case
//     maximum of the one-wave shapes: no difference;
//     deferred solves (the wave parks the moments of 2, 3 or 4 consecutive tiles in LDS, the last tile's in its own dead
//     staging rows, and then solves 32 / 48 / 64 cases at once, one per lane, instead of 16 cases on a quarter of the lanes
//     after every tile — the solve is 26 % of C2's VALU instructions): bit-identical results, same registers and
//     occupancy, but C2 0.183 / 0.229 / 0.189 ms instead of 0.175 and C5 0.381 (2 tiles) / 0.403 (4) instead of 0.352:
//     the short solve after every tile is what the other waves' loads hide behind; VALU instruction count is not the limit.
//     direct xk loads as well (every lane reads the 128 contiguous bytes of its own 8 neighbours with dwordx4 loads, the
//     maximum met by shuffles: no LDS, no barrier at all): 0.189 vs 0.179 ms at three waves per SIMD, 0.241 at
//     __launch_bounds__(64, 3), 0.397 with two lanes per case — 64 distinct lines per load instruction are fine for the
//     fk third of the bytes but not for all of them.
#pragma once
#include <cstdlib>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_moments.hpp"

#ifndef WLSQM_TILE_RUN_MIN_NO
#define WLSQM_TILE_RUN_MIN_NO 6      // A/B: smallest system whose tile stores its fi rows as one run through LDS
#endif
#ifndef WLSQM_TILE_FI_NT
#define WLSQM_TILE_FI_NT 1      // non-temporal stores of the tile's fi run (written once, never read by the kernel): configs[1], interleaved, 0.1523-0.1551 against 0.1550-0.1599 ms (profiles/r03i_ab_tile_nt.txt)
#endif

namespace wlsqm {

constexpr int WV = 64;            // lanes per wave

typedef double double2_ __attribute__((ext_vector_type(2)));   // 16-byte aligned pair -> dwordx4 / ds_*_b128

// round `v` up to the next value that is == r (mod m)
__host__ __device__ constexpr int round_up_mod(int v, int m, int r) { return v + ((r - v % m) % m + m) % m; }

// LDS row strides (in doubles).  xk row: K*DIM doubles, read by lane c at c*RS.
//   DIM == 2: ds_read_b128 of (x,y): RS == 2 (mod 4) makes 16 consecutive lanes hit 16 distinct 4-bank slots.
//   otherwise ds_read_b64: RS odd makes 32 consecutive lanes hit 32 distinct 2-bank slots.
template <int DIM> __host__ __device__ constexpr int row_stride_x(int K) {
    return DIM == 2 ? round_up_mod(K * DIM, 4, 2) : round_up_mod(K * DIM, 2, 1);
}
__host__ __device__ constexpr int row_stride_f(int K) { return round_up_mod(K, 2, 1); }

// K: neighbour slots per row in memory; KC >= K: slots the shares cover (K rounded up so that every share is even;
// the slots beyond K are masked like the unused slots of a ragged case).
template <int DIM, int ORDER, int K, int KSPLIT, int LPC, bool FKD = false, bool MOM = false, int KC = K>
struct TileGeom {
    static constexpr int NO = ndofs(DIM, ORDER);
    static constexpr int NE = NO * (NO + 1) / 2;
    static constexpr int TC = WV / LPC;                      // cases per tile
    static constexpr int NT = WV * KSPLIT;                   // threads per workgroup
    static constexpr int SHARES = KSPLIT * LPC;              // neighbour shares per case
    static constexpr int KPL = KC / SHARES;                  // neighbours per lane
    static constexpr int RS = row_stride_x<DIM>(K), FS = row_stride_f(K);
    static constexpr int XCH = TC * K * DIM / 2, FCH = TC * K / 2;      // 16-byte chunks per tile
    static constexpr int NX = (XCH + NT - 1) / NT, NF = (FCH + NT - 1) / NT;
    static constexpr int CPRX = K * DIM / 2, CPRF = K / 2;  // chunks per row
    static constexpr int NA = MOM ? mom_count<DIM>(2 * ORDER) : NE;   // matrix accumulators: distinct moments or unique entries
    static constexpr int NRED = NA + NO;                     // partial sums per case
    static constexpr int LDS_TILE = TC * (RS + (FKD ? 0 : FS)) + (KC - K) * DIM;   // FKD: fk is read straight from global by its owner lane; the last row's masked slots are read too
    static constexpr int LDS_RED = (KSPLIT - 1) * NRED * TC;
    static constexpr int LDS_MAIN = LDS_TILE > LDS_RED ? LDS_TILE : LDS_RED;
    static constexpr size_t LDS_BYTES = sizeof(double) * (LDS_MAIN + SHARES * TC);
    static_assert((K * DIM) % 2 == 0 && K % 2 == 0, "rows must be multiples of 16 bytes");
    static_assert(KC % SHARES == 0 && KC >= K && KC - K < 16, "the covered slots must split evenly over the shares");
    static_assert(LPC == 1 || LPC == 2 || LPC == 4, "1, 2 or 4 lanes per case");
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
};

// (Interleaved 16-byte pieces of a row for the LPC lanes of a case — one fk load instruction then reads LPC x 16 contiguous bytes per
// row, the change of the k order that was worth 10 % in csrc/solve_op.hip — measured on C2, same box, old / new library: 0.1538-0.1546
// against 0.1529-0.1545 ms: nothing, not kept.)
// UNR: unroll factor of the neighbour loops; MINW: min waves per SIMD for the register allocator
// (__launch_bounds__ 2nd argument).
// GATHER: index-based ("cloud") input — the tile's rows are gathered from the point tables S/F through
// hoods[ncases, K] instead of being read from dense xk/fk; everything after the LDS staging is identical.
// MOM: accumulate the distinct moments (wlsqm_moments.hpp) instead of the matrix entries; wave 0 expands them.
// SPLIT: stop after the reduction and park the moments in the workspace p.ws (fit_moment.hip solves them in a second
//        kernel): for systems whose expanded matrix does not fit the register file next to the accumulators.
template <int DIM, int ORDER, int K, int KSPLIT, int LPC, int UNR, int MINW, bool GATHER, bool FKD = false, bool MOM = false,
          bool SPLIT = false, int KC = K>
__global__ __launch_bounds__(WV * KSPLIT, MINW) void fit_tile_kernel(const KParams p, const long long ntiles_in) {
    const bool no_run_store = (ntiles_in >> 40) & 1;            // A/B switch (WLSQM_HIP_TILE_RUN_STORE=0)
    const long long ntiles = ntiles_in & ((1ll << 40) - 1);
    using G = TileGeom<DIM, ORDER, K, KSPLIT, LPC, FKD, MOM, KC>;
    static_assert(!SPLIT || MOM, "the workspace holds moments");
    static_assert(!(FKD && GATHER), "direct fk loads are a dense-path option");
    static_assert(!FKD || G::KPL % 2 == 0, "direct fk loads need an even share");
    constexpr int NO = G::NO, NE = G::NE, TC = G::TC, NT = G::NT, RS = G::RS, FS = G::FS;
    constexpr int NX = G::NX, NF = G::NF, CPRX = G::CPRX, CPRF = G::CPRF, KPL = G::KPL, NRED = G::NRED, NA = G::NA;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sX = lds;                                      // [TC][RS]
    double* sF = lds + TC * RS;                            // [TC][FS]
    double* sMax = lds + G::LDS_MAIN;                      // [SHARES][TC] partial max_d2 (outside the reused region)

    const int tid = threadIdx.x;
    const int lane = tid & (WV - 1);
    const int wave = tid / WV;                             // wave-uniform
    const int c = lane % TC;                               // case within the tile
    const int h = lane / TC;                               // which of the LPC lanes of that case
    const int share = wave * LPC + h;
    const int k0 = share * KPL;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long j0 = tile * TC;
        const long long j = j0 + c;
        const bool valid = j < p.ncases;
        const long long jc = valid ? j : p.ncases - 1;     // clamp: tail lanes replay the last case, never store
        const long long nvalid = (p.ncases - j0 < TC) ? (p.ncases - j0) : TC;

        // per-case scalars (small, straight to registers)
        const int nkc = min(p.nk[jc * p.snk], K);
        const bool uniform = (p.wm[jc * p.swm] == WLSQM_WEIGHT_UNIFORM);
        unsigned long long known, dropped;
        effective_mask<NO>(p.knowns[jc * p.sknowns], known, dropped);
        double xi[DIM];
        double fdir[FKD ? KPL : 1];     // FKD: this lane's fk values, straight from global memory

        if constexpr (!GATHER) {
            // ---- stage 1: issue every global load of the tile (coalesced 16 B per lane)
            double2_ bx[NX], bf[FKD ? 1 : NF];
            {
                const double2_* gx = reinterpret_cast<const double2_*>(p.xk + j0 * (long long)(K * DIM));
                const double2_* gf = reinterpret_cast<const double2_*>(p.fk + j0 * (long long)K);
                const long long xlim = nvalid * CPRX, flim = nvalid * CPRF;
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const long long q = tid + (long long)i * NT;
                    bx[i] = gx[q < xlim ? q : xlim - 1];
                }
                if constexpr (!FKD) {
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        const long long q = tid + (long long)i * NT;
                        bf[i] = gf[q < flim ? q : flim - 1];
                    }
                } else {
                    // this lane's KPL values of its own case: KPL*8 contiguous bytes of row jc
                    // (a padded share's slots beyond the row replay the row's last pair: masked below)
                    const double* gr = p.fk + jc * (long long)K;
#pragma unroll
                    for (int i = 0; i < KPL / 2; ++i) {
                        const int kq = (KC == K || k0 + 2 * i < K) ? k0 + 2 * i : K - 2;
                        const double2_ v = *reinterpret_cast<const double2_*>(gr + kq);
                        fdir[2 * i] = v.x; fdir[2 * i + 1] = v.y;
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < DIM; ++m) xi[m] = p.xi[jc * p.sxi_j + m];

            // ---- stage 2: park the tile in LDS (padded rows)
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int q = tid + i * NT;
                if (G::XCH % NT == 0 || q < G::XCH) {
                    const int r = q / CPRX, c2 = q - r * CPRX;     // compile-time divisors
                    double* d = sX + r * RS + 2 * c2;
                    if constexpr (RS % 2 == 0) *reinterpret_cast<double2_*>(d) = bx[i];
                    else { d[0] = bx[i].x; d[1] = bx[i].y; }
                }
            }
            if constexpr (!FKD) {
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const int q = tid + i * NT;
                    if (G::FCH % NT == 0 || q < G::FCH) {
                        const int r = q / CPRF, c2 = q - r * CPRF;
                        double* d = sF + r * FS + 2 * c2;
                        d[0] = bf[i].x; d[1] = bf[i].y;
                    }
                }
            }
        } else {
            // ---- stage 1: the tile's neighbour lists, TC*K int32 contiguous (coalesced 16 B per lane) ...
            // (16 B per lane where K is a multiple of 4, 8 B otherwise)
            constexpr int HW = (K % 4 == 0) ? 4 : 2;
            constexpr int HCH = TC * K / HW, NH = (HCH + NT - 1) / NT, CPRH = K / HW;
            typedef int int4_ __attribute__((ext_vector_type(HW)));
            int4_ hb[NH];
            int live_slots[NH];       // slots of chunk i's row that are real neighbours, counted from the chunk's first slot
            {
                const int4_* gh = reinterpret_cast<const int4_*>(p.hoods + j0 * (long long)K);
                const long long hlim = nvalid * CPRH;
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    const long long q = tid + (long long)i * NT;
                    const long long qq = q < hlim ? q : hlim - 1;
                    hb[i] = gh[qq];
                    const int rr = (int)(qq / CPRH);                          // compile-time divisor
                    live_slots[i] = p.nk[(j0 + rr) * p.snk] - HW * (int)(qq - (long long)rr * CPRH);
                }
            }
            const long long pj = own_point(p, jc);
#pragma unroll
            for (int m = 0; m < DIM; ++m) xi[m] = p.S[pj * DIM + m];
            // ... then the gathers of the point rows they name (HW per index chunk), all in flight together.  Slots
            // k >= nk[row] are padding and may hold anything (-1, npoints, ...; the reference never reads them either,
            // simple.pyx:147): they are never dereferenced — the lane reads its own case's point instead, masked later.
            double gx[NH * HW][DIM], gf[NH * HW];
#pragma unroll
            for (int i = 0; i < NH; ++i)
#pragma unroll
                for (int e = 0; e < HW; ++e) {
                    const long long idx = e < live_slots[i] ? (long long)hb[i][e] : pj;
                    if constexpr (DIM == 2) {
                        const double2_ v = *reinterpret_cast<const double2_*>(p.S + idx * 2);
                        gx[i * HW + e][0] = v.x; gx[i * HW + e][1] = v.y;
                    } else {
#pragma unroll
                        for (int m = 0; m < DIM; ++m) gx[i * HW + e][m] = p.S[idx * DIM + m];
                    }
                    gf[i * HW + e] = p.F[idx];
                }
            // ---- stage 2: park them in the same padded LDS image the dense path builds
#pragma unroll
            for (int i = 0; i < NH; ++i) {
                const int q = tid + i * NT;
                if (HCH % NT == 0 || q < HCH) {
                    const int r = q / CPRH, kq = HW * (q - r * CPRH);
#pragma unroll
                    for (int e = 0; e < HW; ++e) {
#pragma unroll
                        for (int m = 0; m < DIM; ++m) sX[r * RS + (kq + e) * DIM + m] = gx[i * HW + e][m];
                        sF[r * FS + kq + e] = gf[i * HW + e];
                    }
                }
            }
        }
        __syncthreads();

        const double* xr = sX + c * RS;
        const double* fr = sF + c * FS;

        // ---- pass 1: largest squared distance of the case (impl.pyx:389-391): every share scans its
        // neighbours, the partial maxima meet in LDS.  Neighbours k >= nk[j] of a ragged case count as 0.
        double max_d2 = 0.0;
#pragma unroll UNR
        for (int kk = 0; kk < KPL; ++kk) {
            const int k = k0 + kk;
            double d2 = 0.0;
#pragma unroll
            for (int m = 0; m < DIM; ++m) { const double dd = xr[k * DIM + m] - xi[m]; d2 += dd * dd; }
            d2 = (k < nkc) ? d2 : 0.0;
            max_d2 = d2 > max_d2 ? d2 : max_d2;
        }
        if constexpr (G::SHARES > 1) {
            sMax[share * TC + c] = max_d2;
            __syncthreads();
#pragma unroll
            for (int s = 0; s < G::SHARES; ++s) { const double o = sMax[s * TC + c]; max_d2 = o > max_d2 ? o : max_d2; }
        }
        const double inv_max = inverse_max(max_d2);

        // ---- pass 2: this lane's share of the neighbours.  A wave whose cases all use the full K
        // neighbours (the common case) runs the loop without the per-neighbour `live` selects.
        double A[NA], g[NO];               // MOM: moments mu / nu (graded order); else: packed upper triangle of M / g
#pragma unroll
        for (int e = 0; e < NA; ++e) A[e] = 0.0;
#pragma unroll
        for (int a = 0; a < NO; ++a) g[a] = 0.0;
        auto neighbour = [&](int k, bool live) {
            double d[DIM];
            if constexpr (DIM == 2) {
                const double2_ xy = *reinterpret_cast<const double2_*>(xr + 2 * k);   // ds_read_b128
                d[0] = xy.x - xi[0]; d[1] = xy.y - xi[1];
            } else {
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = xr[k * DIM + m] - xi[m];
            }
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = live ? d[m] : 0.0;
            const double fv = FKD ? fdir[k - k0] : fr[k];
            const double f = live ? fv : 0.0;
            if constexpr (MOM) {
                double d2 = 0.0;
#pragma unroll
                for (int m = 0; m < DIM; ++m) d2 += d[m] * d[m];
                const double w = live ? weight(d2, inv_max, uniform) : 0.0;
                accumulate_moments_best<DIM, ORDER>(A, g, d, w, f);
            } else {
                double cc[NO];
                const double d2 = monomials<DIM, ORDER>(d, cc);
                const double w = live ? weight(d2, inv_max, uniform) : 0.0;
                accumulate<NO>(A, g, cc, w, f);
            }
        };
        if (KC == K && __all(nkc >= K)) {
#pragma unroll UNR
            for (int kk = 0; kk < KPL; ++kk) neighbour(k0 + kk, true);
        } else {
#pragma unroll 1
            for (int kk = 0; kk < KPL; ++kk) neighbour(k0 + kk, k0 + kk < nkc);
        }

        // ---- sum the LPC lanes of a case (lanes c, c+TC, ...) with wave shuffles
        if constexpr (LPC > 1) {
#pragma unroll
            for (int off = TC; off < WV; off <<= 1) {
#pragma unroll
                for (int e = 0; e < NA; ++e) A[e] += __shfl_xor(A[e], off, WV);
#pragma unroll
                for (int a = 0; a < NO; ++a) g[a] += __shfl_xor(g[a], off, WV);
            }
        }
        // ---- sum the KSPLIT waves through LDS (the tile's storage is dead now)
        if constexpr (KSPLIT > 1) {
            __syncthreads();
            double* red = lds;
            if (wave > 0 && h == 0) {
                double* mine = red + (wave - 1) * (NRED * TC) + c;
#pragma unroll
                for (int e = 0; e < NA; ++e) mine[e * TC] = A[e];
#pragma unroll
                for (int a = 0; a < NO; ++a) mine[(NA + a) * TC] = g[a];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int w = 1; w < KSPLIT; ++w) {
                    const double* other = red + (w - 1) * (NRED * TC) + c;
#pragma unroll
                    for (int e = 0; e < NA; ++e) A[e] += other[e * TC];
#pragma unroll
                    for (int a = 0; a < NO; ++a) g[a] += other[(NA + a) * TC];
                }
            }
        }

        // ---- wave 0: [expand the moments,] knowns elimination, LDL^T, substitution, store
        if constexpr (SPLIT) {
            if (wave == 0 && valid && h == 0) {
                double* w = p.ws + j;
#pragma unroll
                for (int e = 0; e < NA; ++e) w[e * p.ws_stride] = A[e];
#pragma unroll
                for (int a = 0; a < NO; ++a) w[(NA + a) * p.ws_stride] = g[a];
            }
        } else if (wave == 0) {
            constexpr unsigned long long FULL = (1ull << NO) - 1ull;
            // One wave per tile, whole tile, no case without unknowns or with dropped DOFs, contiguous 16-byte aligned fi rows: the tile's TC rows are
            // ONE run of TC no doubles; they go through LDS (the tile's image is dead: a wave's LDS operations complete in order)
            // and leave as 16-byte pieces.  Separate 8-byte stores at a `no`-double pitch are the slow pattern of this memory system
            // (csrc/fit_sens.hip), and the wave waits for their acknowledgement before the next tile's loads can be consumed.
            // 1M C2 cases, interleaved A/B (WLSQM_HIP_TILE_RUN_STORE=0 / default), three runs: 0.171 / 0.173 / 0.174 -> 0.168 / 0.168 /
            // 0.166 ms on a slow box of the pool; without any fi store: -10 %.  Three unknowns (C1: 96 doubles per tile) lose 2 %
            // and keep the direct stores.
            bool run_store = false;
            if constexpr (KSPLIT == 1 && (TC * NO) % 2 == 0 && NO >= WLSQM_TILE_RUN_MIN_NO) {
                run_store = !no_run_store && nvalid == TC && p.sfi_j == NO && __all(dropped == 0ull && known != FULL) &&
                            ((reinterpret_cast<uintptr_t>(p.fi) & 15u) == 0);
            }
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
                    if (run_store) {
                        // (a known DOF is re-written with its own bits, as the reference's Case_get_fi does: infra.pyx:780-795)
#pragma unroll
                        for (int a = 0; a < NO; ++a) lds[c * NO + a] = ((known >> a) & 1ull) ? fio[a] : rhs[a];
                    } else {
#pragma unroll
                        for (int a = 0; a < NO; ++a)
                            if (!((known >> a) & 1ull)) fio[a] = rhs[a];
                    }
                };
                if constexpr (MOM) {
                    double M[NE], rhs[NO];
                    expand_moments<DIM, ORDER>(A, g, M, rhs);
                    finish(M, rhs);
                } else {
                    finish(A, g);
                }
            }
            if constexpr (KSPLIT == 1 && (TC * NO) % 2 == 0 && NO >= WLSQM_TILE_RUN_MIN_NO) {
                if (run_store) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    double2_* out = reinterpret_cast<double2_*>(p.fi + j0 * NO);
                    const double2_* src = reinterpret_cast<const double2_*>(lds);
#pragma unroll
                    for (int q0 = 0; q0 < TC * NO / 2; q0 += WV) {
                        const int q = q0 + lane;
#if WLSQM_TILE_FI_NT
                        if ((TC * NO / 2) % WV == 0 || q < TC * NO / 2) __builtin_nontemporal_store(src[q], &out[q]);
#else
                        if ((TC * NO / 2) % WV == 0 || q < TC * NO / 2) out[q] = src[q];
#endif
                    }
                }
            }
        }
        // the next tile overwrites LDS.  One wave per workgroup: the LDS operations of a wave complete in order, so only the compiler
        // must not reorder them — __syncthreads() would also be `s_waitcnt vmcnt(0)`, i.e. wait for the fi stores just issued to
        // be acknowledged BEFORE the next tile's loads go out (1M C2 cases: 0.139 ms without the stores, 0.155 with them)
        if constexpr (KSPLIT == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    }
}

template <int DIM, int ORDER, int K, int KSPLIT, int LPC = 1, int UNR = 2, int MINW = 2, bool FKD = false, bool MOM = false>
static int launch_tile_any(const KParams& p, hipStream_t stream, bool gather);

template <int DIM, int ORDER, int K, int KSPLIT, int LPC, int UNR, int MINW, bool GATHER, bool FKD = false, bool MOM = false,
          bool SPLIT = false, int KC = K>
static int launch_tile_impl(const KParams& p, hipStream_t stream) {
    using G = TileGeom<DIM, ORDER, K, KSPLIT, LPC, FKD, MOM, KC>;
    constexpr size_t lds_bytes = G::LDS_BYTES;
    const long long ntiles = (p.ncases + G::TC - 1) / G::TC;
    static KernelSetup setup;
    auto kern = fit_tile_kernel<DIM, ORDER, K, KSPLIT, LPC, UNR, MINW, GATHER, FKD, MOM, SPLIT, KC>;
    long long grid = 0;
    int rc = persistent_grid(reinterpret_cast<const void*>(kern), G::NT, lds_bytes, lds_bytes, true, setup, &grid);
    if (rc != WLSQM_OK) return rc;
    if (grid > ntiles) grid = ntiles;
    const char* rs = getenv("WLSQM_HIP_TILE_RUN_STORE");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G::NT), lds_bytes, stream, p, ntiles | ((rs && rs[0] == '0') ? (1ll << 40) : 0));
    WLSQM_HIP_CHECK(hipGetLastError());
    if (!SPLIT) note_kernel(GATHER ? "tile-gather" : "tile");
    return WLSQM_OK;
}

// FKD (fk read straight from global memory) only exists for dense input; the index-based mode stages F[hoods].
template <int DIM, int ORDER, int K, int KSPLIT, int LPC, int UNR, int MINW, bool FKD, bool MOM>
static int launch_tile_any(const KParams& p, hipStream_t stream, bool gather) {
    if (gather) return launch_tile_impl<DIM, ORDER, K, KSPLIT, LPC, UNR, MINW, true, false, MOM>(p, stream);
    return launch_tile_impl<DIM, ORDER, K, KSPLIT, LPC, UNR, MINW, false, FKD, MOM>(p, stream);
}

}  // namespace wlsqm
