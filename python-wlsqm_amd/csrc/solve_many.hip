// solve_many.hip — many right-hand sides on one prepared geometry (BASELINE config 4: "prepare once + 256 RHS solves").
//
// The reference solves one field per ExpertSolver.solve() call (expert.pyx:467-655) and keeps the factored matrices
// between calls.  Here a solve re-fits from the resident coordinates (expert.hip), 852 B per case and field for C2.
// When R fields are available AT ONCE (independent data sets on one geometry; not a time loop, where field t+1 depends
// on the solution for t), the geometry work can be shared: this kernel stages a tile's xk once, builds per lane the
// weighted monomial rows w_k c_k[.] of its neighbours (kept in VGPRs), the normal matrix and its LDL^T factor once,
// and then streams the R right-hand sides: per field and case only fk (8 nk B) is read and fi (8 no B) written —
// 304 B instead of 852 B for C2 — and the arithmetic shrinks to g = (W C)^T f, the knowns correction and a substitution.
//
// Per case this IS the small GEMM (W C)^T [no x nk] x F [nk x R] of the north star's "batched-GEMM solve", but it is
// bound by the fk stream (1.3 flop per byte), and on MI355X fp64 MFMA has the vector rate, so it runs on plain fp64 FMAs
// with the 6 x 8 operator slice of each lane held in registers.
//
// Same one-wave tile shape as fit_tile1_kernel (wlsqm_tile1.hpp).  Register budget limits it to no <= 6 and K <= 32
// (2D order <= 2, 3D order <= 1, 1D order <= 4 ... the time-stepping shapes); anything else falls back to R launches of
// the fused kernel in expert.hip.
#include <algorithm>

#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"
#include "wlsqm_tile1.hpp"

namespace wlsqm {

struct ManyRhs {
    long long nrhs;
    const double* fk; long long sfk_r, sfk_j;      // fk[r * sfk_r + j * sfk_j + k], k contiguous, sfk_j == K
    double* fi;       long long sfi_r, sfi_j;      // fi[r * sfi_r + j * sfi_j + a]
};

template <int DIM, int ORDER, int FMAX, int LPC = K1_LPC>
__global__ __launch_bounds__(K1_WV, 2) void solve_many_kernel(const KParams p, const long long ntiles, const Tile1Geom G,
                                                              const ManyRhs R) {
    constexpr int NO = ndofs(DIM, ORDER), NE = NO * (NO + 1) / 2, TC = K1_WV / LPC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* sX = lds;
    const int lane = threadIdx.x, c = lane % TC, h = lane / TC;
    const int k0 = h * G.KPL;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long j0 = tile * TC, j = j0 + c;
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

        const double* frow = R.fk + jc * R.sfk_j;
        double fnext[FMAX];
        tile1_load_f<FMAX>(fnext, frow, k0, G);                  // field 0, consumed after the geometry work

        tile1_stage_x<DIM>(sX, p.xk + j0 * (long long)(G.K * DIM), nvalid, lane, G);
        __syncthreads();
        const double* xr = sX + c * G.RS;
        const double inv_max = inverse_max(tile1_max_d2<DIM, FMAX, LPC>(xr, xi, k0, nkc, G));

        // ---- geometry, once per tile: this lane's rows of W C, the normal matrix, its factor
        double wc[FMAX][NO], M[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) M[e] = 0.0;
#pragma unroll
        for (int kk = 0; kk < FMAX; ++kk) {
            if (kk < G.KPL) {              // wave-uniform
                const int k = k0 + kk;
                const bool live = k < nkc;
                const int kc = live ? k : 0;
                double d[DIM], cc[NO];
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = xr[kc * DIM + m] - xi[m];
                const double d2 = monomials<DIM, ORDER>(d, cc);
                const double w = live ? weight(d2, inv_max, uniform) : 0.0;
#pragma unroll
                for (int a = 0; a < NO; ++a) wc[kk][a] = (a == 0) ? w : w * cc[a];
#pragma unroll
                for (int a = 0; a < NO; ++a)
#pragma unroll
                    for (int b = a; b < NO; ++b) M[tri<NO>(a, b)] = fma(wc[kk][a], (b == 0) ? 1.0 : cc[b], M[tri<NO>(a, b)]);
            }
        }
        __syncthreads();                   // the tile's rows are dead
#pragma unroll
        for (int off = TC; off < K1_WV; off <<= 1)
#pragma unroll
            for (int e = 0; e < NE; ++e) M[e] += __shfl_xor(M[e], off, K1_WV);
        // unfactored entries: the knowns correction of every right-hand side needs them; parked in LDS (the tile's rows
        // are dead), one copy per case, so that the registers go to the right-hand-side prefetch instead
        double* sMo = lds + c;             // [NE][TC]
        if (h == 0) {
#pragma unroll
            for (int e = 0; e < NE; ++e) sMo[e * TC] = M[e];
        }
        {
            double g0[NO], v0[NO];
#pragma unroll
            for (int a = 0; a < NO; ++a) { g0[a] = 0.0; v0[a] = 0.0; }
            eliminate_knowns<NO>(M, g0, known, v0);               // identity rows/columns for the knowns
        }
        ldlt_factor<NO>(M);

        constexpr unsigned long long FULL = (1ull << NO) - 1ull;
        const bool store = valid && h == 0 && known != FULL;

        // ---- the right-hand sides
        for (long long r = 0; r < R.nrhs; ++r) {
            double f[FMAX];
#pragma unroll
            for (int kk = 0; kk < FMAX; ++kk) f[kk] = fnext[kk];
            // prefetch the next field (two fields ahead measured slower: 0.093 vs 0.077 ms per field, register spills; also
            // when the registers are freed by keeping only (offset, weight) per neighbour and recomputing the monomials: 0.081)
            if (r + 1 < R.nrhs) tile1_load_f<FMAX>(fnext, frow + (r + 1) * R.sfk_r, k0, G);
            double g[NO];
#pragma unroll
            for (int a = 0; a < NO; ++a) g[a] = 0.0;
#pragma unroll
            for (int kk = 0; kk < FMAX; ++kk)
                if (kk < G.KPL) {
                    // slots beyond nk[j] may hold anything (padding of the device rows): 0 * NaN would poison the sum
                    const double fv = (k0 + kk < nkc) ? f[kk] : 0.0;
#pragma unroll
                    for (int a = 0; a < NO; ++a) g[a] = fma(wc[kk][a], fv, g[a]);
                }
#pragma unroll
            for (int off = TC; off < K1_WV; off <<= 1)
#pragma unroll
                for (int a = 0; a < NO; ++a) g[a] += __shfl_xor(g[a], off, K1_WV);
            double* fio = R.fi + r * R.sfi_r + jc * R.sfi_j;
            if (known) {
                // knowns elimination of this field (impl.pyx:815-818): values from its fi row
#pragma unroll
                for (int om = 0; om < NO; ++om) {
                    if ((known >> om) & 1ull) {
                        const double v = (((known & ~dropped) >> om) & 1ull) ? fio[om] : 0.0;
#pragma unroll
                        for (int a = 0; a < NO; ++a)
                            if (a != om) g[a] -= sMo[sym<NO>(a, om) * TC] * v;
                    }
                }
#pragma unroll
                for (int om = 0; om < NO; ++om)
                    if ((known >> om) & 1ull) g[om] = 0.0;
            }
            ldlt_solve<NO>(M, g);
            if (store) {
#pragma unroll
                for (int a = 0; a < NO; ++a)
                    if (!((known >> a) & 1ull)) fio[a] = g[a];
            }
        }
        __syncthreads();                   // the next tile overwrites LDS (sMo)
    }
}

template <int DIM, int ORDER, int FMAX = 8, int LPC = K1_LPC>
static int launch_many(const KParams& p, long long K, const ManyRhs& R, hipStream_t stream, bool* handled) {
    constexpr int K1_TC = K1_WV / LPC;          // (shadows the default tile size)
    Tile1Geom G;
    if (!tile1_geometry<DIM, LPC>(K, FMAX, G)) return WLSQM_OK;
    constexpr int NE_ = ndofs(DIM, ORDER) * (ndofs(DIM, ORDER) + 1) / 2;
    const size_t lds_bytes = sizeof(double) * (size_t)std::max(K1_TC * G.RS, K1_TC * NE_);   // tile rows, then the unfactored matrices
    *handled = true;
    const long long ntiles = (p.ncases + K1_TC - 1) / K1_TC;
    auto kern = solve_many_kernel<DIM, ORDER, FMAX, LPC>;
    static KernelSetup setup;
    long long grid = 0;
    int rc = persistent_grid(reinterpret_cast<const void*>(kern), K1_WV, lds_bytes, 0, false, setup, &grid);
    if (rc != WLSQM_OK) return rc;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(K1_WV), lds_bytes, stream, p, ntiles, G, R);
    WLSQM_HIP_CHECK(hipGetLastError());
    note_kernel("solve-many");
    return WLSQM_OK;
}

// R right-hand sides on the dense resident geometry in p (p.fk / p.fi unused).  `handled` stays false when the shape
// has no instantiation (no > 6, K > 32, K*dim odd, strided geometry, bucketed orders): the caller falls back to R
// fused launches.
int launch_solve_many(int dimension, int order, const KParams& p, long long K, long long nrhs,
                      const double* fk, long long sfk_r, long long sfk_j, double* fi, long long sfi_r, long long sfi_j,
                      hipStream_t stream, bool* handled) {
    *handled = false;
    if (p.case_index || p.hoods || p.do_sens || p.iterative) return WLSQM_OK;
    if (K < 4 || ((K * dimension) % 2) != 0 || sfk_j != K) return WLSQM_OK;
    if (p.sxk_k != dimension || p.sxk_j != K * dimension) return WLSQM_OK;
    if ((reinterpret_cast<uintptr_t>(p.xk) | reinterpret_cast<uintptr_t>(fk)) & 15u) return WLSQM_OK;
    if ((sfk_r % 2) != 0) return WLSQM_OK;                          // every field's rows stay 16-byte aligned
    const ManyRhs R{nrhs, fk, sfk_r, sfk_j, fi, sfi_r, sfi_j};
    // (two lanes per case with 16 neighbours each — LPC 2, FMAX 16 — spills the 96-double operator slice: C4 16.5 instead of 4.8 ms;
    // compiled for a lone wave per SIMD it does not spill and comes to 5.15 against 4.91 ms in the same run: close, not better)
#define MCASE(D, O) if (dimension == D && order == O) return launch_many<D, O>(p, K, R, stream, handled);
    MCASE(1, 0) MCASE(1, 1) MCASE(1, 2) MCASE(1, 3) MCASE(1, 4)
    MCASE(2, 0) MCASE(2, 1) MCASE(2, 2)
    MCASE(3, 0) MCASE(3, 1)
#undef MCASE
    return WLSQM_OK;
}

}  // namespace wlsqm
