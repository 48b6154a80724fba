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
//      1000 C3 cases against the lane kernel's 5.7 and 2.7-3.0 here.
//
// 200k cases, ms per launch, generic kernels (lane per case; wave per case for 3D orders 3 / 4) -> this path: 2D order 4 at
// K = 50 / 26: 1.14 / 0.59 -> 0.87 / 0.57; 2D order 3 at K = 80: 1.31 -> 0.71; 2D order 2 at K = 160: 2.70 -> 0.74; 3D order 2 at K = 160:
// 3.02 -> 1.42; 1D order 2 at K = 100: 1.42 -> 0.36; 3D order 3 at K = 60: 4.05 -> 2.59; 3D order 4 at K = 100: 22.7 -> 7.9
// (tools/time_sens.py, profiles/r02e_time_sens.txt).
//
// The sensitivities are a linear map of the right-hand side (sens[k, :] = A^-1 (w c[k, :]), known DOFs masked), so the explicit
// inverse gives the same numbers as the substitutions up to rounding of the order cond(A) eps — the bound every other path has.
// The batch is cut into slices so that the scratch for the inverses stays bounded (200k cases, slices of 32 / 96 / 256 / 1024 MB:
// C3-like 1.27 / 1.07 / 1.05 / 1.04 ms, 3D order 4 at K = 100 22.3 / 20.3 / 19.2 / 18.7 ms with the first versions of the kernels
// — no Infinity Cache effect to be had).
#include "wlsqm_internal.hpp"
#include "wlsqm_kernels.hpp"

namespace wlsqm {

bool chunk_inverse_ok(int dimension, int order, const KParams& p, long long K);
int launch_fit_chunk_inverse(int dimension, int order, const KParams& p, long long K, hipStream_t stream);
int launch_fit_wave_inverse(int dimension, int order, const KParams& p, double* inv, hipStream_t stream);
int launch_fit_rows_inverse(int dimension, int order, const KParams& p, double* inv, hipStream_t stream);
bool moment_inverse_ok(int dimension, int order, const KParams& p, long long max_nk);
int launch_fit_moment_inverse(int dimension, int order, const KParams& p, long long max_nk, double* inv, hipStream_t stream);

typedef double sd4_ __attribute__((ext_vector_type(4)));

// One case at a time per wave, no LDS, 41 registers for 15 unknowns (8 waves per SIMD hide the latencies).  Every lane of a
// 16-neighbour block computes the monomials of ITS neighbour itself (the four lanes of a neighbour redundantly: ~60 VALU
// instructions against an LDS round trip and two wave syncs) and stores its four results — DOFs 4 v + g of neighbour n — straight
// from the MFMA accumulator: 32-byte pieces, four instructions covering the block's contiguous 16 no doubles.
// Measured alternatives, 200k C3-like cases (2D order 4, K = 50), this kernel 0.54-0.60 ms (2.0-2.2 TB/s of output; PMC: 500 VALU +
// 233 SALU + 16 MFMA instructions per case, SIMDs 53 % busy, 55 % of the wave cycles in s_waitcnt, HBM writes = the output):
//   - weighted monomial rows computed once per 64 neighbours and passed to the blocks through LDS, results staged in LDS and
//     stored as 512-byte runs, persistent waves with the next case prefetched: 0.54 ms (201 registers, two waves per SIMD);
//     capped at 128 registers it spills: 1.04 ms;
//   - the same with direct stores (LDS only for the monomial rows, 86 registers): 0.54 ms, no change — and this kernel with its
//     stores removed still takes 55 % (15 unknowns) to 78 % (2D order 2, K = 160) of its time: neither the redundant VALU work
//     nor the store pattern is what bounds it, the dependent chain load -> reduce -> weights -> MFMA -> store of ONE case per
//     wave is (8 waves per SIMD is the hardware's limit);
//   - one workgroup per case instead of waves walking the batch: the same time;
//   - workgroups sharing an L2 (blockIdx % 8) walking a contiguous eighth of the batch: 43 % less fetched (the 120-byte runs of
//     two neighbouring cases share lines), 5 % slower.
template <int DIM, int ORDER>
__global__ __launch_bounds__(64) void sens_apply_kernel(const KParams p, const double* __restrict__ inv_all, const int grouped) {
    constexpr int WV = 64, NO = ndofs(DIM, ORDER);
    constexpr int RB = (NO + 15) / 16, KS = (NO + 3) / 4;
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
#pragma unroll 1
    for (int kb = 0; kb < nkc; kb += WV) {
        if (kb > 0) {
            const int kk = kb + lane < kmax ? kb + lane : kmax - 1;
#pragma unroll
            for (int m = 0; m < DIM; ++m) x0[m] = xr[kk * p.sxk_k + m];
        }
#pragma unroll 1
        for (int nb = 0; nb < 4; ++nb) {
            const int k0 = kb + nb * 16;
            if (k0 >= nkc) break;                                          // wave-uniform
            const int k = k0 + n;
            const bool live = k < nkc;
            double B[KS];
            {
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
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                sd4_ acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[rb][s], B[s], acc, 0, 0, 0);
                // D[4 v + l / 16][l % 16]: DOF a = 16 rb + 4 v + g of neighbour n; NaN for knowns (impl.pyx:821-823)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int a = 16 * rb + 4 * v + g;
                    if (live && (16 * rb + 4 * v + 3 < NO || a < NO)) {
                        if (!((known >> a) & 1ull)) row[16 * rb + 4 * v] = acc[v];
                        else if (!((dropped >> a) & 1ull)) row[16 * rb + 4 * v] = qnan;
                    }
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
// WLSQM_HIP_SENS_SLICE_MB sets the size of a slice's inverses (default 256).
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
    const double slice_mb = (mb && atof(mb) > 0.0) ? atof(mb) : 256.0;
    long long per = (long long)(slice_mb * 1048576.0 / (8.0 * no * no));
    per = per < 1024 ? 1024 : per;
    per = (per / 64) * 64;                                                 // whole tiles and whole groups of 64 cases
    if (per > p.ncases) per = p.ncases;
    double* inv = nullptr;
    int rc = scratch_alloc_async(reinterpret_cast<void**>(&inv), (size_t)((per + 63) / 64 * 64) * no * no * sizeof(double), stream);
    if (rc != WLSQM_OK) return rc;
    for (long long j0 = 0; j0 < p.ncases && rc == WLSQM_OK; j0 += per) {
        const long long n = (p.ncases - j0 < per) ? (p.ncases - j0) : per;
        KParams q = slice_cases(p, j0, n);
        q.max_nk = K;
        if (big) rc = wave_form ? launch_fit_wave_inverse(dimension, order, q, inv, stream) : launch_fit_rows_inverse(dimension, order, q, inv, stream);
        else if (mom) rc = launch_fit_moment_inverse(dimension, order, q, K, inv, stream);
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
