// fit_sens.hip — the fit with sensitivities (do_sens, impl.pyx:776-778, 821-846) for the shapes without a tile kernel of their
// own: 15-unknown systems (2D order 4), any K > 128 (K > 64 for the small systems), 3D orders 3 and 4.  Per slice of the batch:
//
//   1. the basic fit, which also leaves the INVERSE of every case's knowns-eliminated normal matrix in scratch memory: `no`
//      substitutions with unit vectors per case instead of one substitution per neighbour (impl.pyx:831-834 runs dgetrs over
//      all nk right-hand sides).  2D order 4 up to K = 100: the two-kernel moment path (fit_moment.hip; its solve kernel has one
//      case per lane and the factor in registers); other systems up to 15 unknowns: the chunked any-K kernel (fit_chunk.hip);
//      3D orders 3 / 4: the row-per-lane kernel (fit_rows.hip; WLSQM_HIP_SENS_WAVE=1: the LDS form of fit_wave.hip, A/B);
//   2. sens_apply_kernel: one case at a time per wave, the lanes rebuild the neighbours' weighted monomial rows w c[k, :]
//      (impl.pyx:826-829) and the wave multiplies the rows of 16 neighbours with the inverse on the matrix cores
//      (v_mfma_f64_16x16x4, the inverse as the A operand, loaded once per case with ordinary vector loads).  A per-case matrix
//      has to reach all of the case's neighbour lanes; the MFMA does that broadcast in hardware.  Measured alternative: the
//      inverse as SGPR operands of v_fma_f64 (scalar loads, the case is wave-uniform) — every row of the inverse is a
//      scalar-cache miss the wave waits out (15 serialised ~1 us round trips per case, 100 SGPRs hold two rows): 3.7 us per
//      1000 C3 cases against the lane kernel's 5.7 and 2.1 here.
//
// 200k cases, ms per launch, generic kernels (lane per case; wave per case for 3D orders 3 / 4) -> this path: 2D order 4 at
// K = 50 / 26: 1.14 / 0.59 -> 0.72 / 0.55; 2D order 3 at K = 80: 1.31 -> 0.61; 2D order 2 at K = 160: 2.70 -> 0.78; 3D order 2 at K = 160:
// 3.02 -> 1.15; 1D order 2 at K = 100: 1.42 -> 0.32; 3D order 3 at K = 60: 4.05 -> 2.39; 3D order 4 at K = 100: 22.7 -> 7.7
// (tools/time_sens.py, profiles/r02e_time_sens.txt).
//
// The sensitivities are a linear map of the right-hand side (sens[k, :] = A^-1 (w c[k, :]), known DOFs masked), so the explicit
// inverse gives the same numbers as the substitutions up to rounding of the order cond(A) eps — the bound every other path has.
// The batch is cut into slices so that the scratch for the inverses stays bounded (200k cases, slices of 32 / 96 / 256 / 1024 MB:
// C3-like 1.27 / 1.07 / 1.05 / 1.04 ms, 3D order 4 at K = 100 22.3 / 20.3 / 19.2 / 18.7 ms with the first versions of the kernels
// — no Infinity Cache effect to be had).
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

#ifndef WLSQM_SENS_NT
#define WLSQM_SENS_NT 1      // non-temporal stores of the sensitivities' 16-byte pieces (written once, never read here): 400k cases of 2D order 4, K = 50: 1.39-1.42 against 1.45-1.46 ms, K = 64: 1.54-1.56 against 1.56-1.60 (profiles/r03i_ab_sens_nt.txt)
#endif

namespace wlsqm {

bool chunk_inverse_ok(int dimension, int order, const KParams& p, long long K);
int launch_fit_chunk_inverse(int dimension, int order, const KParams& p, long long K, hipStream_t stream);
int launch_fit_wave_inverse(int dimension, int order, const KParams& p, double* inv, hipStream_t stream);
int launch_fit_rows_inverse(int dimension, int order, const KParams& p, double* inv, hipStream_t stream);
bool moment_inverse_ok(int dimension, int order, const KParams& p, long long max_nk);
int launch_fit_moment_inverse(int dimension, int order, const KParams& p, long long max_nk, double* inv, hipStream_t stream);
int launch_fit_stage_inverse(int dimension, int order, const KParams& p, long long K, double* inv, hipStream_t stream, bool* handled);      // fit_stage.hip

typedef double sd4_ __attribute__((ext_vector_type(4)));

// One case at a time per wave, waves walking the batch; 16 neighbours per MFMA block; results of the systems with >= 10 unknowns
// leave as 16-byte pieces of the block's contiguous run (through a 16-row LDS image), the small systems store straight from
// the accumulator.  What was measured on the way, 200k C3-like cases (2D order 4, K = 50), ms per launch of this kernel
// (100k cases each):
//   - every lane computing the monomials of its block's neighbour (no LDS at all, 41 registers, 8 waves per SIMD), 8-byte
//     stores of the accumulator's 32-byte pieces at a 120-byte pitch: 0.29-0.30 — and 0.31 with the arithmetic REMOVED, 0.16
//     with the stores removed: that store pattern runs at 1.9 TB/s whatever is computed beside it (HBM writes = the output,
//     no read-modify-write; PMC: 500 VALU + 233 SALU + 16 MFMA instructions per case);
//   - the same with the block's run stored as 16-byte pieces through a 16 x no LDS image: 0.23; with the monomial rows computed
//     once per 64 neighbours and handed to the blocks through LDS (the image in place of the block's rows): 0.21 (2.7 TB/s);
//   - rows and images in LDS, results staged for the whole 64-neighbour group, persistent waves with the next case
//     prefetched (201 registers, two waves per SIMD): 0.27; capped at 128 registers it spills: 0.52;
//   - one workgroup per case instead of waves walking the batch: the same time;
//   - workgroups sharing an L2 (blockIdx % 8) walking a contiguous eighth of the batch: 43 % less fetched (the 120-byte runs of
//     two neighbouring cases share lines), 5 % slower;
//   - a coordinate load per 16-neighbour block (behind the previous block's stores: vmcnt is one in-order queue): no change
//     once the store pattern was the bound, kept out anyway.
// LDS traffic between the lanes of ONE wave (one wave per workgroup): LDS operations of a wave complete in order, so only the
// compiler has to be kept from moving a read above the write it depends on — no s_barrier, and no vmcnt(0) (which
// __syncthreads() implies and which would wait out the stores in flight).
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef double sd2_ __attribute__((ext_vector_type(2)));

template <int DIM, int ORDER>
__global__ __launch_bounds__(64) void sens_apply_kernel(const KParams p, const double* __restrict__ inv_all, const int grouped) {
    constexpr int WV = 64, NO = ndofs(DIM, ORDER);
    constexpr int RB = (NO + 15) / 16, KS = (NO + 3) / 4;
    // From 10 unknowns on a block's results leave through LDS as 16-byte pieces of its contiguous run (200k cases, 15 unknowns,
    // K = 50: 0.29 -> 0.23 ms; 3D order 2 at K = 160: 1.42 -> 1.20 ms).  The small systems are bound by their arithmetic
    // (2D order 2 at K = 160: 0.46 of 0.60 ms without any store) and lose with the staging (0.60 -> 0.61, four blocks per image: 0.66).
    constexpr bool STAGE = NO >= 10;
    constexpr int IMG = 1;                                                 // 16-neighbour blocks per output image
    // Up to 15 unknowns the weighted monomial rows of a 64-neighbour group are computed ONCE, one neighbour per lane, and handed
    // to the blocks through LDS (row stride `no`); a block's rows are dead once its B operand has been read, and its output
    // image takes their place.  (Without it every lane computes the row of its block's neighbour, four lanes the same one:
    // the SIMD pays 4 x ~85 VALU instructions per 64 neighbours instead of 85 + 15 LDS writes + 4 x 4 reads.)
    constexpr bool SHARE = NO <= 15;
    __shared__ __attribute__((aligned(16))) double sO[SHARE ? WV * NO : (STAGE ? IMG * 16 * NO : 2)];
    constexpr unsigned long long FULL = (NO >= 64) ? ~0ull : ((1ull << NO) - 1ull);
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int kmax = (int)p.max_nk;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  for (long long t = blockIdx.x; t < p.ncases; t += gridDim.x) {
    const int nkc = min(p.nk[t * p.snk], kmax);
    const bool uniform = (p.wm[t * p.swm] == WLSQM_WEIGHT_UNIFORM);
    unsigned long long known, dropped;
    effective_mask<NO>(p.knowns[t * p.sknowns], known, dropped);
    if (known == FULL) continue;                                           // nr < 1: no-op (impl.pyx:574, 636, 742)
    double xi[DIM];
#pragma unroll
    for (int m = 0; m < DIM; ++m) xi[m] = p.xi[t * p.sxi_j + m];
    const double* xr = p.xk + t * p.sxk_j;
    // A operand (lane l: A[l % 16][l / 16]): rows 16 rb + n, columns 4 s + g of the (symmetric) inverse
    const double* inv = grouped ? inv_all + (t >> 6) * (long long)(64 * NO * NO) + (t & 63) * NO : inv_all + t * (long long)(NO * NO);
    const int cs = grouped ? 64 * NO : NO;
    double A[RB][KS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int i = 16 * rb + n, b = 4 * s + g;
            A[rb][s] = (i < NO && b < NO) ? inv[b * cs + i] : 0.0;
        }
    // Neighbour kb + lane's coordinates, 64 at a time, BEFORE any store of the group: vmcnt counts loads and stores in one
    // in-order queue, so a load issued behind a block's stores makes the wave wait out their acknowledgement (measured with
    // one load per 16-neighbour block: 55 % of the wave cycles in s_waitcnt, 19 us per case).  The blocks fetch their
    // neighbour from the owning lane (ds_bpermute: LDS crossbar, not vmcnt).
    double x0[DIM];
    {
        const int kk = lane < kmax ? lane : kmax - 1;
#pragma unroll
        for (int m = 0; m < DIM; ++m) x0[m] = xr[kk * p.sxk_k + m];
    }
    double max_d2 = 0.0;
    if (!uniform) {
        if (lane < nkc) {
#pragma unroll
            for (int m = 0; m < DIM; ++m) { const double dd = x0[m] - xi[m]; max_d2 = fma(dd, dd, max_d2); }
        }
        for (int k = lane + WV; k < nkc; k += WV) {
            double d2 = 0.0;
#pragma unroll
            for (int m = 0; m < DIM; ++m) { const double dd = xr[k * p.sxk_k + m] - xi[m]; d2 = fma(dd, dd, d2); }
            max_d2 = d2 > max_d2 ? d2 : max_d2;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(max_d2, off, WV); max_d2 = o > max_d2 ? o : max_d2; }
    }
    const double inv_max = inverse_max(max_d2);
    double* out = p.sens + t * p.ss_j + g;
    // rows of exactly `no` doubles, every DOF written, 16-byte aligned blocks: the block's 16 no consecutive doubles leave as
    // 16-byte pieces through LDS (wave-uniform)
    const bool runs = STAGE && p.ss_k == NO && dropped == 0 && ((reinterpret_cast<uintptr_t>(p.sens + t * p.ss_j)) & 15u) == 0;
#pragma unroll 1
    for (int kb = 0; kb < nkc; kb += WV) {
        if (kb > 0) {
            const int kk = kb + lane < kmax ? kb + lane : kmax - 1;
#pragma unroll
            for (int m = 0; m < DIM; ++m) x0[m] = xr[kk * p.sxk_k + m];
        }
        if constexpr (SHARE) {
            lds_wave_sync();                                               // the previous group's rows and images have been read
            double d[DIM], c[NO];
#pragma unroll
            for (int m = 0; m < DIM; ++m) d[m] = x0[m] - xi[m];            // (rows past nk: computed, not stored)
            const double d2 = monomials<DIM, ORDER>(d, c);
            const double w = weight(d2, inv_max, uniform);
#pragma unroll
            for (int b = 0; b < NO; ++b) sO[lane * NO + b] = (b == 0) ? w : w * c[b];
            lds_wave_sync();
        }
#pragma unroll 1
        for (int nb = 0; nb < 4; ++nb) {
            const int k0 = kb + nb * 16;
            if (k0 >= nkc) break;                                          // wave-uniform
            const int k = k0 + n;
            const bool live = k < nkc;
            double* img = SHARE ? sO + nb * 16 * NO : sO;                  // this block's rows / output image
            double B[KS];
            if constexpr (SHARE) {
                // B operand (lane l: B[l / 16][l % 16]): entry 4 s + g of neighbour n's row
#pragma unroll
                for (int s = 0; s < KS; ++s) B[s] = (4 * s + 3 < NO || 4 * s + g < NO) ? img[n * NO + 4 * s + g] : 0.0;
            } else {
                double d[DIM], c[NO];
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = __shfl(x0[m], nb * 16 + n, WV) - xi[m];      // (rows past nk: computed, not stored)
                const double d2 = monomials<DIM, ORDER>(d, c);
                const double w = weight(d2, inv_max, uniform);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double c0 = (4 * s == 0) ? 1.0 : c[4 * s < NO ? 4 * s : 0];
                    const double c1 = (4 * s + 1 < NO) ? c[4 * s + 1 < NO ? 4 * s + 1 : 0] : 0.0;
                    const double c2 = (4 * s + 2 < NO) ? c[4 * s + 2 < NO ? 4 * s + 2 : 0] : 0.0;
                    const double c3 = (4 * s + 3 < NO) ? c[4 * s + 3 < NO ? 4 * s + 3 : 0] : 0.0;
                    B[s] = w * ((g == 0) ? c0 : (g == 1) ? c1 : (g == 2) ? c2 : c3);
                }
            }
            double* row = out + (long long)k * p.ss_k;
            const int ib = SHARE ? 0 : nb % IMG;                           // block of the image
            if (runs && ib == 0) lds_wave_sync();                          // the previous image / this block's rows have been read
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                sd4_ acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[rb][s], B[s], acc, 0, 0, 0);
                // D[4 v + l / 16][l % 16]: DOF a = 16 rb + 4 v + g of neighbour n; NaN for knowns (impl.pyx:821-823)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int a = 16 * rb + 4 * v + g;
                    if (16 * rb + 4 * v + 3 < NO || a < NO) {
                        if (runs) img[(ib * 16 + n) * NO + a] = ((known >> a) & 1ull) ? qnan : acc[v];
                        else if (live) {
                            if (!((known >> a) & 1ull)) row[16 * rb + 4 * v] = acc[v];
                            else if (!((dropped >> a) & 1ull)) row[16 * rb + 4 * v] = qnan;
                        }
                    }
                }
            }
            if (runs && (ib == IMG - 1 || k0 + 16 >= nkc)) {
                // Separate 8-byte stores of 32-byte pieces at a 120-byte pitch (the accumulator layout) run at 1.9 TB/s whatever
                // is computed beside them (measured with the arithmetic removed); whole 16-byte pieces of the image's run do not.
                lds_wave_sync();
                const int k_img = k0 - ib * 16;                            // first neighbour of the image
                const int total = min(IMG * 16, nkc - k_img) * NO;         // doubles of its run
                double* run = p.sens + t * p.ss_j + (long long)k_img * NO;
#pragma unroll
                for (int q0 = 0; q0 < IMG * 16 * NO / 2; q0 += WV) {
                    const int q = q0 + lane;
#if WLSQM_SENS_NT
                    if (2 * q + 1 < total) __builtin_nontemporal_store(*reinterpret_cast<const sd2_*>(img + 2 * q), reinterpret_cast<sd2_*>(run + 2 * q));
#else
                    if (2 * q + 1 < total) *reinterpret_cast<sd2_*>(run + 2 * q) = *reinterpret_cast<const sd2_*>(img + 2 * q);
#endif
                    else if (2 * q < total) run[2 * q] = img[2 * q];
                }
            }
        }
    }
  }
}

// Iterative refinement (impl.pyx:986-1083) on the same inverse: one case at a time per wave, ONE LANE PER NEIGHBOUR.  A sweep
// evaluates the model at the neighbours (the coefficient vector is wave-uniform, read from LDS), takes the max-norm of the
// residual with the reference's semantics (the first residual seeds the maximum, so a NaN there poisons it, impl.pyx:1037-1041),
// sums the lanes' shares of C^T W res through LDS (lane a adds column a: 64 conflict-free reads instead of 6 no shuffles) and
// lane a forms its component of the correction from row a of the inverse, which it keeps in registers for the whole case.
// The sweeps stop when the norm repeats exactly (impl.pyx:1057), as every refinement path of this library does.
template <int DIM, int ORDER>
__global__ __launch_bounds__(64) void refine_apply_kernel(const KParams p, const double* __restrict__ inv_all, const int grouped) {
    constexpr int WV = 64, NO = ndofs(DIM, ORDER), NOP = NO | 1;
    constexpr unsigned long long FULL = (NO >= 64) ? ~0ull : ((1ull << NO) - 1ull);
    __shared__ double sR[WV * NOP];
    __shared__ double sF[NO + 1];
    const int lane = threadIdx.x, me = lane < NO ? lane : NO - 1;
    const int kmax = (int)p.max_nk;
    int iters_all = 0;
    for (long long t = blockIdx.x; t < p.ncases; t += gridDim.x) {
        const int nkc = min(p.nk[t * p.snk], kmax);
        const bool uniform = (p.wm[t * p.swm] == WLSQM_WEIGHT_UNIFORM);
        unsigned long long known, dropped;
        effective_mask<NO>(p.knowns[t * p.sknowns], known, dropped);
        if (known == FULL) continue;                                       // nr < 1: no-op (impl.pyx:574, 636, 742)
        double xi[DIM];
#pragma unroll
        for (int m = 0; m < DIM; ++m) xi[m] = p.xi[t * p.sxi_j + m];
        const double* xr = p.xk + t * p.sxk_j;
        const double* fr = p.fk + t * p.sfk_j;
        double* fio = p.fi + t * p.sfi_j;
        // row `lane` of the (symmetric) inverse; rows and columns of known DOFs are zero there
        double Arow[NO];
        {
            const double* inv = grouped ? inv_all + (t >> 6) * (long long)(64 * NO * NO) + (t & 63) * NO : inv_all + t * (long long)(NO * NO);
            const int cs = grouped ? 64 * NO : NO;
#pragma unroll
            for (int b = 0; b < NO; ++b) Arow[b] = inv[b * cs + me];
        }
        double myfi = fio[me];                                             // the basic fit's solution; knowns hold the user's values
        double max_d2 = 0.0;
        if (!uniform) {
            for (int k = lane; k < nkc; k += WV) {
                double d2 = 0.0;
#pragma unroll
                for (int m = 0; m < DIM; ++m) { const double dd = xr[k * p.sxk_k + m] - xi[m]; d2 = fma(dd, dd, d2); }
                max_d2 = d2 > max_d2 ? d2 : max_d2;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(max_d2, off, WV); max_d2 = o > max_d2 ? o : max_d2; }
        }
        const double inv_max = inverse_max(max_d2);

        double prev_norm = -1.0;
        bool broke = false;
        int i = 0;
        for (i = 0; i < p.max_iter; ++i) {
            __syncthreads();
            if (lane < NO) sF[lane] = myfi;
            __syncthreads();
            double fi[NO], racc[NO];
#pragma unroll
            for (int a = 0; a < NO; ++a) { fi[a] = sF[a]; racc[a] = 0.0; }
            double norm = 0.0;
            bool first = true;
            for (int kb = 0; kb < nkc; kb += WV) {
                const int k = kb + lane;
                const bool live = k < nkc;
                const int kc = live ? k : kb;
                double d[DIM], c[NO];
#pragma unroll
                for (int m = 0; m < DIM; ++m) d[m] = xr[kc * p.sxk_k + m] - xi[m];
                const double d2 = monomials<DIM, ORDER>(d, c);
                const double w = weight(d2, inv_max, uniform);
                double model = fi[0];                                      // taylor_*D (polyeval.pyx): sum_a c[a] fi[a]
#pragma unroll
                for (int a = 1; a < NO; ++a) model += c[a] * fi[a];
                const double res = live ? fr[kc * p.sfk_k] - model : 0.0;
                const double ar = live ? fabs(res) : -1.0;                 // lanes without a neighbour never win the maximum
                const double wr = w * res;
#pragma unroll
                for (int a = 0; a < NO; ++a) racc[a] += (a == 0) ? wr : wr * c[a];
                const double ar0 = __shfl(ar, 0, WV);
                double cm = (ar == ar) ? ar : -1.0;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) { const double o = __shfl_xor(cm, off, WV); cm = o > cm ? o : cm; }
                if (first) { norm = (ar0 == ar0) ? cm : ar0; first = false; }
                else if (cm > norm) norm = cm;
            }
            if (norm == prev_norm) { broke = true; break; }                 // impl.pyx:1057 (wave-uniform)
            prev_norm = norm;
#pragma unroll
            for (int a = 0; a < NO; ++a) sR[lane * NOP + a] = racc[a];
            __syncthreads();
            double r = 0.0;
            if (lane < NO) {
#pragma unroll 8
                for (int l = 0; l < WV; ++l) r += sR[l * NOP + lane];
            }
            __syncthreads();
            if (lane < NO) sF[lane] = r;
            __syncthreads();
            double corr = 0.0;
#pragma unroll
            for (int b = 0; b < NO; ++b) corr = fma(Arow[b], sF[b], corr);
            if (!((known >> me) & 1ull)) myfi += corr;
        }
        const int iters = broke ? i : (p.max_iter > 0 ? p.max_iter : 1);    // for/else, impl.pyx:1080-1081
        if (lane < NO && !((known >> me) & 1ull)) fio[lane] = myfi;
        iters_all = iters > iters_all ? iters : iters_all;
    }
    if (lane == 0 && p.iters_out && iters_all > 0) atomicMax(p.iters_out, iters_all);
}

template <int DIM, int ORDER>
static int launch_refine(const KParams& p, const double* inv, bool grouped, hipStream_t stream) {
    static KernelSetup setup;
    auto kern = refine_apply_kernel<DIM, ORDER>;
    long long grid = 0;
    int rc = persistent_grid(reinterpret_cast<const void*>(kern), 64, 0, 0, true, setup, &grid);
    if (rc != WLSQM_OK) return rc;
    grid = (long long)((double)grid / grid_multiple() * 2.0);
    if (grid < 1) grid = 1;
    if (grid > p.ncases) grid = p.ncases;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), 0, stream, p, inv, grouped ? 1 : 0);
    WLSQM_HIP_CHECK(hipGetLastError());
    return WLSQM_OK;
}

static int refine_dispatch(int dimension, int order, const KParams& p, const double* inv, bool grouped, hipStream_t stream) {
#define ACASE(D, O) if (dimension == D && order == O) return launch_refine<D, O>(p, inv, grouped, stream);
    ACASE(1, 0) ACASE(1, 1) ACASE(1, 2) ACASE(1, 3) ACASE(1, 4)
    ACASE(2, 0) ACASE(2, 1) ACASE(2, 2) ACASE(2, 3) ACASE(2, 4)
    ACASE(3, 0) ACASE(3, 1) ACASE(3, 2) ACASE(3, 3) ACASE(3, 4)
#undef ACASE
    set_error("refine_apply: unsupported (dimension, order)");
    return WLSQM_EVALUE;
}

// Two workgroups per resident slot walk the batch (WLSQM_HIP_SENS_GRID_MULT: 1 / 2 / 4 / 8 measured within 3 %).
template <int DIM, int ORDER>
static int launch_apply(const KParams& p, const double* inv, bool grouped, hipStream_t stream) {
    static KernelSetup setup;
    auto kern = sens_apply_kernel<DIM, ORDER>;
    long long grid = 0;
    int rc = persistent_grid(reinterpret_cast<const void*>(kern), 64, 0, 0, true, setup, &grid);
    if (rc != WLSQM_OK) return rc;
    const char* e = getenv("WLSQM_HIP_SENS_GRID_MULT");
    const double mult = (e && atof(e) > 0.0) ? atof(e) : 2.0;
    grid = (long long)((double)grid / grid_multiple() * mult);
    if (grid < 1) grid = 1;
    if (grid > p.ncases) grid = p.ncases;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), 0, stream, p, inv, grouped ? 1 : 0);
    WLSQM_HIP_CHECK(hipGetLastError());
    return WLSQM_OK;
}

static int apply_dispatch(int dimension, int order, const KParams& p, const double* inv, bool grouped, hipStream_t stream) {
#define ACASE(D, O) if (dimension == D && order == O) return launch_apply<D, O>(p, inv, grouped, stream);
    ACASE(1, 0) ACASE(1, 1) ACASE(1, 2) ACASE(1, 3) ACASE(1, 4)
    ACASE(2, 0) ACASE(2, 1) ACASE(2, 2) ACASE(2, 3) ACASE(2, 4)
    ACASE(3, 0) ACASE(3, 1) ACASE(3, 2) ACASE(3, 3) ACASE(3, 4)
#undef ACASE
    set_error("sens_apply: unsupported (dimension, order)");
    return WLSQM_EVALUE;
}

// Dense input (any strides for the 3D order-3/4 systems; contiguous even-K rows for the others — api.hip repacks), sensitivities
// without refinement.  WLSQM_HIP_DISABLE_SENS_APPLY=1 leaves these shapes to the generic kernels (A/B);
// WLSQM_HIP_SENS_SLICE_MB sets the size of a slice's inverses (default 1024).
int launch_fit_sens(int dimension, int order, const KParams& p, long long K, hipStream_t stream, bool* handled) {
    *handled = false;
    const char* off = getenv("WLSQM_HIP_DISABLE_SENS_APPLY");
    const char* off2 = getenv("WLSQM_HIP_DISABLE_TILE");
    if ((off && off[0] == '1') || (off2 && off2[0] == '1')) return WLSQM_OK;
    const bool want_sens = p.do_sens && p.sens;
    if ((!want_sens && !p.iterative) || p.hoods || p.case_index || !p.xk || !p.fk || K < 1) return WLSQM_OK;
    const int no = wlsqm_hip_number_of_dofs(dimension, order);
    const bool big = no > 15;
    // Refinement: a wave per case pays per-case overheads (syncs, the 64-row LDS sums) the lane-per-case kernel does not have, and
    // that kernel's weakness — strided row reads — matters less when the rows are re-read from cache sweep after sweep.  200k
    // cases, generic kernel -> inverse + refine_apply_kernel, ms: 2D order 4 at K = 50 / 26: 1.20 / 0.66 -> 2.16 / 2.23, 2D order 3 at
    // K = 80: 1.64 -> 1.70 (those stay on the generic kernel); 2D order 2 at K = 160: 5.12 -> 1.32, 3D order 2 at K = 160: 3.68 -> 2.03,
    // 1D order 2 at K = 100: 2.94 -> 0.81, 3D order 3 at K = 60: 6.64 -> 3.74, 3D order 4 at K = 100: 24.6 -> 10.9.
    if (p.iterative && !(big || K > 128 || no <= 6)) return WLSQM_OK;
    const char* nomom = getenv("WLSQM_HIP_SENS_NO_MOMENT");               // A/B: the chunked kernel for 2D order 4 too
    const bool mom = !big && !(nomom && nomom[0] == '1') && moment_inverse_ok(dimension, order, p, K);
    if (big ? !(dimension == 3 && (order == 3 || order == 4)) : (!mom && !chunk_inverse_ok(dimension, order, p, K))) return WLSQM_OK;
    if (p.ncases > 0x7fffffffLL) return WLSQM_OK;
    *handled = true;
    const char* wf = getenv("WLSQM_HIP_SENS_WAVE");                       // A/B: the LDS form of fit_wave.hip for the 3D order-3/4 inverses
    const bool wave_form = wf && wf[0] == '1';
    const char* mb = getenv("WLSQM_HIP_SENS_SLICE_MB");
    // (round 5: 1 GB instead of 256 MB — 400k configs[2] cases in one slice instead of three: 1.335 -> 1.264 ms; the launches and the
    // workspace of every slice cost more than the Infinity Cache residency of a small slice's inverses gains)
    const double slice_mb = (mb && atof(mb) > 0.0) ? atof(mb) : 1024.0;
    long long per = (long long)(slice_mb * 1048576.0 / (8.0 * no * no));
    per = per < 1024 ? 1024 : per;
    per = (per / 64) * 64;                                                 // whole tiles and whole groups of 64 cases
    if (per > p.ncases) per = p.ncases;
    double* inv = nullptr;
    int rc = scratch_alloc_async(reinterpret_cast<void**>(&inv), (size_t)((per + 63) / 64 * 64) * no * no * sizeof(double), stream);
    // (a card short of memory gets smaller slices, not an error — down to 16 MB of inverses)
    while (rc != WLSQM_OK && per > 1024 && (size_t)per * no * no * sizeof(double) > (16u << 20)) {
        (void)hipGetLastError();
        per = ((per / 2) / 64) * 64;
        if (per < 1024) per = 1024;
        rc = scratch_alloc_async(reinterpret_cast<void**>(&inv), (size_t)((per + 63) / 64 * 64) * no * no * sizeof(double), stream);
    }
    if (rc != WLSQM_OK) return rc;
    for (long long j0 = 0; j0 < p.ncases && rc == WLSQM_OK; j0 += per) {
        const long long n = (p.ncases - j0 < per) ? (p.ncases - j0) : per;
        KParams q = slice_cases(p, j0, n);
        q.max_nk = K;
        if (big) rc = wave_form ? launch_fit_wave_inverse(dimension, order, q, inv, stream) : launch_fit_rows_inverse(dimension, order, q, inv, stream);
        else if (mom) {
            bool staged = false;                                      // (round 5: the staged one-lane-per-case fit writes the inverse itself where the input is its kind)
            rc = launch_fit_stage_inverse(dimension, order, q, K, inv, stream, &staged);
            if (rc == WLSQM_OK && !staged) rc = launch_fit_moment_inverse(dimension, order, q, K, inv, stream);
        }
        else {
            KParams a = q;
            a.ws = inv; a.do_sens = 0; a.sens = nullptr; a.iterative = 0;
            rc = launch_fit_chunk_inverse(dimension, order, a, K, stream);
        }
        if (rc == WLSQM_OK && want_sens) rc = apply_dispatch(dimension, order, q, inv, mom, stream);
        if (rc == WLSQM_OK && p.iterative) rc = refine_dispatch(dimension, order, q, inv, mom, stream);
    }
    const int rc2 = scratch_free_async(inv, stream);
    if (rc == WLSQM_OK) note_kernel(p.iterative ? (want_sens ? "sens-refine-apply" : "refine-apply") : "sens-apply");
    return rc != WLSQM_OK ? rc : rc2;
}

}  // namespace wlsqm
