// fit_rows.hip — one wavefront per case for the large 3D systems (order 3: 20 unknowns, order 4: 35), basic fit.
//
// The no x no normal matrix does not fit one lane, so the 64 lanes of a wave share a case — but unlike fit_wave.hip
// (the general version, which also does sensitivities and refinement and keeps the matrix in LDS) everything that is
// touched repeatedly lives in registers and crosses lanes as wave-uniform broadcasts (v_readlane), not through LDS:
//   1. lane k loads neighbour k, computes its weight and the power tables w dx^n, dy^n, dz^n (n <= 2*order) into LDS;
//   2. ASSEMBLY IN MOMENT FORM (wlsqm_moments.hpp): the no(no+1)/2 = 630 matrix entries of order 4 are only 165 distinct
//      moments sum_k w dx^p dy^q dz^r; they are dealt to the lanes (<= 3 each), every lane sums its moments over the
//      neighbours with three table look-ups and two multiplies per moment and neighbour (one table row spans fewer than
//      64 banks: conflict-free whatever the lanes pick); lane a < no also sums the right-hand-side moment of DOF a;
//   3. ONE MATRIX ROW PER LANE IN REGISTERS: lane i expands row i from the moments (one pass over LDS), knowns become
//      identity rows/columns (the mask is wave-uniform: one case per wave);
//   4. left-looking LDL^T with the rows in place: at column c every lane i >= c updates v_i = A[i][c] - sum_m V_i[m]
//      (V_c[m] / d_m), where V_c[m] comes from lane c by v_readlane — 2 readlanes + 2 fp64 operations per (c, m), no LDS,
//      no barrier; 1/d_c is computed once per column by all lanes;
//   5. forward substitution the same way; for the backward one the factor is parked in LDS once (column access).
// Per case ~5.5k wave-instructions for order 4 with 64 neighbours, against ~25k (and ~100 LDS round trips on the critical
// path of its rolled loops) in fit_wave.hip.  Order 3 has only 20 rows, so three of its cases share a wave (template
// parameters G = 3 cases x GS = 21 lanes); the pivot row then travels through a per-group LDS mailbox instead of v_readlane.
// Measured (200k cases, 64 neighbours): order 4 7.9 -> 2.2 ms, order 3 2.3 -> 1.2 (one case per wave) -> 0.99 ms; the LDS
// mailbox form with one case per wave is 10 % slower than v_readlane for both orders, the bpermute form with three cases
// per wave 15 % slower than the mailbox.  Sensitivities / refinement stay in fit_wave.hip: grafting its extras onto this kernel's
// front end was measured (order 4 do_sens 8.6 vs 7.3 ms per 100k cases): their 64 LDS substitutions per case dominate and
// the extra LDS state costs occupancy.
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_moments.hpp"

namespace wlsqm {

constexpr int RW = 64;

__device__ __forceinline__ double lane_bcast(double v, int l) {          // value of lane l (compile-time l), wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double inv_factorial(int n) { return n <= 1 ? 1.0 : n == 2 ? 0.5 : n == 3 ? 1.0 / 6.0 : 1.0 / 24.0; }

// G cases per wave, GS lanes each (G * GS <= 64): order 4 fills a wave with one case (35 rows), order 3 (20 rows) would
// leave two thirds of the lanes idle in the factorisation, so three cases share a wave there (GS = 21).  With G > 1 the
// pivot row comes from the group's own lane (ds_bpermute instead of v_readlane) and the neighbours go through LDS in
// chunks of 32 per case.
// INV (G == 1): besides fi, the wave leaves the inverse of the case's eliminated normal matrix at p.ws[t][no][no] for the
// sensitivities (fit_sens.hip).  Lane c substitutes unit vector c through the factor that sits one row per lane in registers:
// L[i][j] reaches all lanes as a wave-uniform value (two v_readlane + one fma per term, 2 x no (no - 1) / 2 terms), no LDS; the
// 64 right-hand sides of fit_wave.hip's LDS form cost three LDS operations per term.
template <int DIM, int ORDER, int G, int GS, bool INV = false>
__global__ __launch_bounds__(RW) void fit_rows_kernel(const KParams p) {
    static_assert(!INV || G == 1, "the inverse is written by the one-case-per-wave form");
    static_assert(DIM == 3, "row-per-lane kernel is instantiated for the 3D systems");
    constexpr int NO = ndofs(DIM, ORDER), NM = mom_count<DIM>(2 * ORDER), NP = 2 * ORDER + 1;
    constexpr int TS = DIM * NP, TSP = TS | 1;                    // power-table row: [w dx^n | dy^n | dz^n], odd stride
    constexpr int MPL = (NM + GS - 1) / GS;                       // moments per lane
    constexpr int CH = G == 1 ? RW : 32;                          // neighbours of a case per LDS chunk
    constexpr int ROUNDS = (CH + GS - 1) / GS;                    // table rows each lane builds per chunk
    static_assert(G * GS <= RW && NO <= GS && TS <= 32, "one row per lane; one table row inside 64 banks");
    __shared__ double sP[G * CH * TSP];                           // power tables of a chunk; later the factor
    __shared__ double sF[G * CH];
    __shared__ double sMu[G * NM];
    __shared__ double sNu[G * NO];
    __shared__ double sRed[RW];

    const int lane = threadIdx.x;
    const bool idle = lane >= G * GS;                             // G = 3: lane 63
    const int grp = idle ? G - 1 : lane / GS, li = idle ? GS - 1 : lane - grp * GS, gbase = grp * GS;
    const long long t = (long long)blockIdx.x * G + grp;
    const long long ncases = live_cases(p);
    if ((long long)blockIdx.x * G >= ncases) return;             // (block-uniform: before any barrier)
    const bool valid = t < ncases && !idle;
    const long long tc = t < ncases ? t : ncases - 1;
    const long long j = p.case_index ? p.case_index[tc] : tc;
    const int nk = min(p.nk[j * p.snk], (int)p.max_nk);      // never past the end of a row
    const bool uniform = (p.wm[j * p.swm] == WLSQM_WEIGHT_UNIFORM);
    unsigned long long known, dropped;
    effective_mask<NO>(p.knowns[j * p.sknowns], known, dropped);
    constexpr unsigned long long FULL = (1ull << NO) - 1ull;
    if (G == 1 && known == FULL) return;                          // nr < 1: no-op (impl.pyx:574, 636, 742); whole wave exits
    const bool store = valid && known != FULL;

    // dense rows (xk / fk with strides) or index-based: rows hoods[j, k] of the point tables S / F
    const int* hr = p.hoods ? p.hoods + j * p.shoods_j : nullptr;
    const double* xr = hr ? nullptr : p.xk + j * p.sxk_j;
    const double* fr = hr ? nullptr : p.fk + j * p.sfk_j;
    auto coord = [&](int k, int m) { return hr ? p.S[(long long)hr[k] * DIM + m] : xr[k * p.sxk_k + m]; };
    auto value = [&](int k) { return hr ? p.F[hr[k]] : fr[k * p.sfk_k]; };
    double xi[DIM];
    {
        const long long pj = hr ? (own_point(p, j)) : 0;
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = hr ? p.S[pj * DIM + m] : p.xi[j * p.sxi_j + m];
    }
    double* fio = p.fi + j * p.sfi_j;

    // ---- per-lane bookkeeping: table offsets of this lane's moments and of its DOF
    int off[MPL][DIM];
#pragma unroll
    for (int i = 0; i < MPL; ++i) {
        int e = li + GS * i;
        if (e >= NM) e = NM - 1;                                  // clamp: harmless duplicate, never stored
        int s = 0;
        while (mtet(s + 1) <= e) ++s;                             // total degree
        int rem = e - mtet(s), tq = 0;
        while (mtri(tq + 1) <= rem) ++tq;                         // q + r
        const int r = rem - mtri(tq), q = tq - r, pp = s - tq;
        off[i][0] = pp; off[i][1] = NP + q; off[i][2] = 2 * NP + r;
    }
    const int me = li < NO ? li : NO - 1;
    const int pi = Mono<DIM>::P[me], qi = Mono<DIM>::Q[me], ri = Mono<DIM>::R[me];

    // ---- pass 1: largest squared distance (impl.pyx:389-391) over the case's neighbours
    double max_d2 = 0.0;
    for (int k = li; k < nk; k += GS) {
        double d2 = 0.0;
#pragma unroll
        for (int m = 0; m < DIM; ++m) { const double dd = coord(k, m) - xi[m]; d2 += dd * dd; }
        max_d2 = d2 > max_d2 ? d2 : max_d2;
    }
    int nk_all = nk;                                              // largest neighbour count of the wave's cases
    if constexpr (G == 1) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const double v = __shfl_xor(max_d2, o, RW); max_d2 = v > max_d2 ? v : max_d2; }
    } else {
        sRed[lane] = idle ? 0.0 : max_d2;
        __syncthreads();
        for (int o = 0; o < GS; ++o) { const double v = sRed[gbase + o]; max_d2 = v > max_d2 ? v : max_d2; }
#pragma unroll
        for (int gg = 0; gg < G; ++gg) nk_all = max(nk_all, __shfl(nk, gg * GS, RW));
    }
    const double inv_max = inverse_max(max_d2);

    // ---- pass 2: moments, chunks of CH neighbours per case
    double acc[MPL], nacc = 0.0;
#pragma unroll
    for (int i = 0; i < MPL; ++i) acc[i] = 0.0;
    for (int kb = 0; kb < nk_all; kb += CH) {
        const int kc_all = min(CH, nk_all - kb);                  // rows the loop below walks (wave-uniform)
        __syncthreads();                                          // previous chunk fully consumed
#pragma unroll
        for (int rr = 0; rr < ROUNDS; ++rr) {
            const int kl = li + rr * GS;                          // row of the chunk this lane builds
            if (kl < CH && !idle) {
                const int k = kb + kl;
                const bool live = k < nk;
                double d[DIM], d2 = 0.0;
#pragma unroll
                for (int m = 0; m < DIM; ++m) { d[m] = live ? coord(k, m) - xi[m] : 0.0; d2 += d[m] * d[m]; }
                double* row = sP + (grp * CH + kl) * TSP;
                double v = live ? weight(d2, inv_max, uniform) : 0.0;      // rows beyond nk[j] carry weight 0
#pragma unroll
                for (int m = 0; m < DIM; ++m) {
                    if (m > 0) v = 1.0;
#pragma unroll
                    for (int n = 0; n < NP; ++n) { row[m * NP + n] = v; v *= d[m]; }
                }
                sF[grp * CH + kl] = live ? value(k) : 0.0;
            }
        }
        __syncthreads();
        const double* tab = sP + grp * CH * TSP;
        const double* fv = sF + grp * CH;
#pragma unroll 4
        for (int k = 0; k < kc_all; ++k, tab += TSP) {
#pragma unroll
            for (int i = 0; i < MPL; ++i) acc[i] = fma(tab[off[i][0]] * tab[off[i][1]], tab[off[i][2]], acc[i]);
            nacc = fma((tab[pi] * tab[NP + qi]) * tab[2 * NP + ri], fv[k], nacc);
        }
    }
#pragma unroll
    for (int i = 0; i < MPL; ++i)
        if (li + GS * i < NM && !idle) sMu[grp * NM + li + GS * i] = acc[i];
    if (li < NO && !idle) sNu[grp * NO + li] = nacc;
    __syncthreads();

    // ---- row i of the normal matrix and of the right-hand side, from the moments (factorial constants as impl.pyx:331-349)
    const double fi_ = inv_factorial(pi) * inv_factorial(qi) * inv_factorial(ri);
    double V[NO];
#pragma unroll
    for (int b = 0; b < NO; ++b) {
        const int pb = Mono<DIM>::P[b], qb = Mono<DIM>::Q[b], rb = Mono<DIM>::R[b];
        const double fb = mom_inv_fact(pb) * mom_inv_fact(qb) * mom_inv_fact(rb);
        const int s = pi + qi + ri + pb + qb + rb, tq = qi + ri + qb + rb, r = ri + rb;
        V[b] = sMu[grp * NM + mtet(s) + mtri(tq) + r] * (fi_ * fb);
    }
    double g = sNu[grp * NO + me] * fi_;

    // ---- knowns (impl.pyx:792-818); the mask is per case (wave-uniform when G == 1)
    const bool kn_me = (known >> me) & 1ull;
    if (known) {
#pragma unroll
        for (int b = 0; b < NO; ++b)
            if ((known >> b) & 1ull) {
                const double val = ((dropped >> b) & 1ull) ? 0.0 : fio[b];
                g -= V[b] * val;
            }
#pragma unroll
        for (int b = 0; b < NO; ++b)
            if (kn_me || ((known >> b) & 1ull)) V[b] = (b == me) ? 1.0 : 0.0;
        if (kn_me) g = 0.0;
    }

    if constexpr (G == 1) {
        // ---- left-looking LDL^T, row i in lane i: V[m] = L[i][m] d_m below the diagonal, V[i] = d_i; the pivot row arrives by
        // v_readlane (wave-uniform), no LDS on the critical path.  (The LDS-mailbox form below is 10 % slower here.)
        // (round 3: the pivot row is broadcast ALREADY divided — U[m] = V[m] / d_m = L[i][m], one multiply per lane and column —
        // instead of dividing the broadcast value in every (column, term) pair: 3 instead of 4 instructions per pair,
        // ~580 of the 4 780 vector instructions of an order-4 case; profiles/r03_rows_sq.txt)
        double dinv[NO], U[NO], my_dinv = 1.0;
#pragma unroll
        for (int c = 0; c < NO; ++c) {
            double v = V[c];
#pragma unroll
            for (int m = 0; m < c; ++m) v = fma(-V[m], lane_bcast(U[m], c), v);
            V[c] = v;
            dinv[c] = recip(lane_bcast(v, c));
            U[c] = v * dinv[c];
            my_dinv = (lane == c) ? dinv[c] : my_dinv;
        }
        // ---- forward substitution and the diagonal
#pragma unroll
        for (int m = 0; m < NO; ++m) {
            const double tm = lane_bcast(g, m) * dinv[m];
            g = (lane > m) ? fma(-V[m], tm, g) : g;
        }
        g *= my_dinv;
        // ---- backward substitution: the factor by columns through LDS (the tables are dead)
        constexpr int LDV = NO + 2;
        static_assert(NO * LDV <= CH * TSP, "the factor reuses the table storage");
        __syncthreads();
        double* sV = sP;
        if (lane < NO) {
#pragma unroll
            for (int m = 0; m < NO; ++m) sV[lane * LDV + m] = V[m];
        }
        __syncthreads();
#pragma unroll
        for (int jj = NO - 1; jj >= 1; --jj) {
            const double xj = lane_bcast(g, jj);
            const double l = sV[jj * LDV + me] * my_dinv;        // L[jj][i] = V_jj[i] / d_i
            g = (lane < jj) ? fma(-l, xj, g) : g;
        }
        if constexpr (INV) {
            // L[i][m] = V_i[m] / d_m in place (m < i; the entries from the diagonal on are not read again)
#pragma unroll
            for (int m = 0; m < NO; ++m) V[m] *= dinv[m];
            const bool col = lane < NO && !((known >> me) & 1ull);           // a known DOF: zero row and column
            double X[NO];
#pragma unroll
            for (int i = 0; i < NO; ++i) X[i] = (i == lane && col) ? 1.0 : 0.0;
#pragma unroll
            for (int jj = 0; jj < NO - 1; ++jj)
#pragma unroll
                for (int i = jj + 1; i < NO; ++i) X[i] = fma(-lane_bcast(V[jj], i), X[jj], X[i]);
#pragma unroll
            for (int jj = NO - 1; jj >= 0; --jj) {
                double v = X[jj] * dinv[jj];
#pragma unroll
                for (int i = jj + 1; i < NO; ++i) v = fma(-lane_bcast(V[jj], i), X[i], v);
                X[jj] = v;
            }
            // lane c holds column c = row c; stored by rows so that a store instruction writes `no` consecutive doubles
            if (lane < NO && valid) {
                double* wi = p.ws + t * (long long)(NO * NO) + lane;
#pragma unroll
                for (int i = 0; i < NO; ++i) wi[i * NO] = X[i];
            }
        }
    } else {
        // ---- left-looking LDL^T, row i in lane i of the group.  Registers keep V[m] = L[i][m] d_m; the scaled entries
        // T[i][m] = L[i][m] are published row by row in LDS as they become final (the tables are dead), so that column c reads
        // the pivot row T[c][0..c) with group-uniform (broadcast) LDS reads: one fma per (column, term), no cross-lane shuffles.
        // One wave per workgroup: LDS operations of a wave complete in order, so a row written at column m is visible at
        // every later column; wave_barrier() only keeps the compiler from moving an LDS read above the write it depends on.
        constexpr int LDT = NO + 2 + (NO & 1);                        // even: 16-byte aligned rows
        static_assert(G * NO * LDT <= G * CH * TSP, "the factor reuses the table storage");
        __syncthreads();
        double* sT = sP + grp * NO * LDT;
        double* sPiv = sRed + gbase;                                  // the group's mailbox: pivot / substitution values
        const bool row = li < NO && !idle;
        double my_dinv = 1.0;
#pragma unroll
        for (int c = 0; c < NO; ++c) {
            double v = V[c];
#pragma unroll
            for (int m = 0; m < c; ++m) v = fma(-V[m], sT[c * LDT + m], v);
            V[c] = v;
            if (row && li == c) *sPiv = v;
            __builtin_amdgcn_wave_barrier();
            const double dc = recip(*sPiv);
            my_dinv = (li == c) ? dc : my_dinv;
            if (row) sT[li * LDT + c] = v * dc;                       // only rows below the diagonal are ever read
            __builtin_amdgcn_wave_barrier();
        }
        // ---- forward substitution fused with the diagonal: z = D^-1 L^-1 g
#pragma unroll
        for (int m = 0; m < NO; ++m) {
            if (row && li == m) *sPiv = g * my_dinv;                  // z_m, final once the rows above are folded in
            __builtin_amdgcn_wave_barrier();
            const double zm = *sPiv;
            g = (li > m) ? fma(-V[m], zm, g) : g;
            __builtin_amdgcn_wave_barrier();
        }
        g *= my_dinv;
        // ---- backward substitution: x_i = z_i - sum_{j > i} L[j][i] x_j, column i of T by lane i
#pragma unroll
        for (int jj = NO - 1; jj >= 1; --jj) {
            if (row && li == jj) *sPiv = g;
            __builtin_amdgcn_wave_barrier();
            const double xj = *sPiv;
            g = (li < jj) ? fma(-sT[jj * LDT + me], xj, g) : g;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (store && li < NO && !kn_me) fio[li] = g;
}

template <int DIM, int ORDER, int G, int GS>
static int launch_rows(const KParams& p, hipStream_t stream) {
    const long long blocks = (p.ncases + G - 1) / G;
    if (blocks > 0x7fffffffLL) { set_error("too many cases for one launch"); return WLSQM_EVALUE; }
    hipLaunchKernelGGL((fit_rows_kernel<DIM, ORDER, G, GS>), dim3((unsigned)blocks), dim3(RW), 0, stream, p);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("rows");
    return WLSQM_OK;
}

// The basic fit of one slice, which also leaves every case's inverse normal matrix at inv[t][no][no] (first kernel of fit_sens.hip).
int launch_fit_rows_inverse(int dimension, int order, const KParams& p_in, double* inv, hipStream_t stream) {
    KParams p = p_in;
    p.ws = inv; p.do_sens = 0; p.sens = nullptr; p.iterative = 0; p.case_index = nullptr;
    if (p.ncases > 0x7fffffffLL) { set_error("too many cases for one launch"); return WLSQM_EVALUE; }
    if (dimension == 3 && order == 3) hipLaunchKernelGGL((fit_rows_kernel<3, 3, 1, 64, true>), dim3((unsigned)p.ncases), dim3(RW), 0, stream, p);
    else if (dimension == 3 && order == 4) hipLaunchKernelGGL((fit_rows_kernel<3, 4, 1, 64, true>), dim3((unsigned)p.ncases), dim3(RW), 0, stream, p);
    else { set_error("fit_rows_inverse: unsupported (dimension, order)"); return WLSQM_EVALUE; }
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("rows-inverse");
    return WLSQM_OK;
}

// Basic fit (no sensitivities, no refinement) of the 3D order-3/4 systems on dense (possibly strided) or index-based input.
int launch_fit_rows(int dimension, int order, const KParams& p, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* off = getenv("WLSQM_HIP_DISABLE_ROWS");          // A/B against fit_wave.hip
    if (off && off[0] == '1') return WLSQM_OK;
    if (p.do_sens || p.iterative) return WLSQM_OK;
    if (dimension == 3 && order == 3) {
        *handled = true;
        const char* one = getenv("WLSQM_ROWS_ONE_CASE");         // A/B: one case per wave for order 3 too
        if (one && one[0] == '1') return launch_rows<3, 3, 1, 64>(p, stream);
        return launch_rows<3, 3, 3, 21>(p, stream);
    }
    if (dimension == 3 && order == 4) { *handled = true; return launch_rows<3, 4, 1, 64>(p, stream); }
    return WLSQM_OK;
}

}  // namespace wlsqm
